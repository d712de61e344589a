"""GPU unit tests of the inline-asm building blocks of the hand-scheduled recursion kernel: scalar row
loads, EXEC-masked update and accumulate steps, against their definition in numpy."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def bits(w):
    return np.array([(int(w) >> i) & 1 for i in range(64)], dtype=bool)


@pytest.mark.parametrize("skm,skc", [(2 ** 64 - 1, 2 ** 64 - 1), (0, 2 ** 64 - 1), (2 ** 64 - 1, 0)])
def test_asm_blocks(skm, skc):
    import relate_amd  # noqa: F401  (loads torch's HIP runtime first)
    relate_amd.lib()
    lib = C.CDLL(os.path.join(HERE, "probe", "libasm_probe.so"))
    rng = np.random.default_rng(3)
    words = rng.integers(0, 2 ** 63, 8, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 8, dtype=np.uint64)
    x = rng.uniform(0.1, 2.0, (4, 64))
    out = np.zeros((16, 64))
    wout = np.zeros(8, dtype=np.uint64)
    rx, rho, add_mis, add_all = 0.37, 1e-3, 0.011, 0.023
    kw, fw, uw = 1, 2, 3
    ms, mt = (2 ** 64 - 1) ^ (1 << 17), (1 << 20) - 1
    u64 = C.c_uint64
    rc = lib.probe_run(words.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                       wout.ctypes.data_as(C.c_void_p), C.c_double(rx), C.c_double(rho), C.c_double(add_mis),
                       C.c_double(add_all), u64(skm), u64(skc), C.c_uint32(kw), C.c_uint32(fw), C.c_uint32(uw), u64(ms),
                       u64(mt))
    assert rc == 0
    assert np.array_equal(wout, words)
    skm_b, skc_b = bits(skm), bits(skc)
    mis = [skm_b & ~bits(words[i]) for i in range(4)]          # mismatch lanes of the "earlier" row
    # forward update
    a = [(x[i] + rx) * np.where(mis[i], rho, 1.0) for i in range(4)]
    for i in range(4):
        assert np.array_equal(out[i], a[i]), f"fwd word {i}"
    # validity of words 0..3: word kw uses ms, word fw uses mt, words >= uw hold nothing
    vm = [np.ones(64, bool), bits(ms), bits(mt), np.zeros(64, bool)]
    acc = np.zeros(64)
    for i in range(4):
        acc = acc + np.where(vm[i], a[i], 0.0)
    assert np.array_equal(out[4], acc)
    # backward update
    b = [((x[i] + np.where(mis[i], add_mis, 0.0)) + add_all) * np.where(mis[i], rho, 1.0) for i in range(4)]
    for i in range(4):
        assert np.array_equal(out[5 + i], b[i]), f"bwd word {i}"
    misc = [skc_b & ~bits(words[4 + i]) for i in range(4)]     # mismatch lanes of the current row
    am0 = np.where(~misc[0], b[0], 0) + np.where(~misc[2], b[2], 0)
    am1 = np.where(~misc[1], b[1], 0) + np.where(~misc[3], b[3], 0)
    ax0 = np.where(misc[0], b[0], 0) + np.where(misc[2], b[2], 0)
    ax1 = np.where(misc[1], b[1], 0) + np.where(misc[3], b[3], 0)
    assert np.array_equal(out[9], am0) and np.array_equal(out[10], am1)
    assert np.array_equal(out[11], ax0) and np.array_equal(out[12], ax1)
    ams = np.zeros(64)
    axs = np.zeros(64)
    for i in range(4):
        ams = ams + np.where(vm[i] & ~misc[i], b[i], 0.0)
        axs = axs + np.where(vm[i] & misc[i], b[i], 0.0)
    assert np.array_equal(out[13], ams) and np.array_equal(out[14], axs)
    assert np.allclose(out[15], x[0].sum(), rtol=1e-14) and np.all(out[15] == out[15][0])
