"""A chunk loaded from .haps / .sample / map as bit rows against the same chunk loaded as '0'/'1' bytes: the
packed matrices on the device (rows, their transpose) are the same bytes and the records are the same records,
which are also the oracle's."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_parity import compare_records
from test_haps_loader import make_case

pytestmark = pytest.mark.gpu


def device_bytes(torch, ptr, nbytes):
    """nbytes at device address ptr as a numpy array (hipMemcpy of the runtime torch has loaded)."""
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.restype = C.c_int
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty(nbytes, dtype=np.uint8)
    torch.cuda.synchronize()
    assert hip.hipMemcpy(out.ctypes.data, ptr, nbytes, 2) == 0  # hipMemcpyDeviceToHost
    return out


@pytest.mark.parametrize("N,L,rho", [(130, 700, 1.0), (64, 300, 0.5), (5008, 120, 1.0)])
def test_chunk_from_bit_rows_equals_chunk_from_bytes(tmp_path, N, L, rho):
    import torch
    import relate_amd as ra
    from relate_amd.haps import Haps, row_stride_words
    # one chunk needs max_chunk_size = budget / N > L; the windows then close every dozen SNPs or so
    seq, bp, map_bp, map_cm, (haps, sample, gmap, _), mem = make_case(tmp_path, N, L, seed=N + L, budget=N * (L + 1.0))
    with Haps(haps, sample, gmap, mem) as h:
        assert h.num_chunks == 1
        first, lc, wb = h.chunk(0)
        _, _, _, r = h.positions()
        a = ra.Plan(0)
        h.load_chunk(a, 0, rho=rho)
    b = ra.Plan(0)
    b.load_host(seq, r * rho, wb)
    (pa, na), (pb, nb) = a.packed(), b.packed()
    assert na == nb == ra.lib().relate_packed_bytes(N, L)
    # rows with their slack, then the N columns of ceil(L / 64) words (the 64 words after them are never written)
    stride = row_stride_words(N)
    used = na - 64 * 8
    assert used == (L * stride + 8 * 80 + 64) * 8 + N * ((L + 63) // 64) * 8
    assert np.array_equal(device_bytes(torch, pa, used), device_bytes(torch, pb, used))
    R, W = ra.record_bytes(N), len(wb) - 1
    k0, k1 = (0, N) if N <= 200 else (N - 12, N)
    outs = []
    for p in (a, b):
        out = torch.empty(W * (k1 - k0) * R, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        p.prepare(0.001, 0.001, k0, k1, st)
        p.paint(out.data_ptr(), st)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy().reshape(W, k1 - k0, R))
        p.close()
    assert np.array_equal(outs[0], outs[1])
    want = ol.oracle().paint_stones(seq, r * rho, 0.001, 0.001, wb, k0, k1)
    assert compare_records(outs[0], want, N, f"N={N} from bit rows") == 0.0


def test_config1_200_haplotypes_5mb_from_haps_files(tmp_path):
    """The shape of the reference's own CPU-runnable case (200 haplotypes, 5 Mb, mu = 1.25e-8, Ne = 30 000: about
    4 Ne mu L H(199) = 44 000 SNPs, flat 1 cM/Mb map) written as .haps / .sample / .map, read by relate_haps_*,
    painted from the bit rows, all 200 targets against the oracle."""
    import torch
    import relate_amd as ra
    from relate_amd.haps import Haps
    from test_haps_loader import write_data_set
    N, L = 200, 44000
    rng = np.random.default_rng(1)
    seq, _ = ra.synth_chunk(N, L, 1)
    gaps = rng.geometric(L / 5.0e6, L)  # mean spacing 5 Mb / L
    bp = np.cumsum(gaps).astype(np.int64)
    map_bp = np.linspace(0, int(bp[-1]) + 1000, 50).astype(np.int64)
    map_cm = map_bp * 1e-6  # 1 cM per Mb
    haps, sample, gmap, _ = write_data_set(str(tmp_path), seq, bp, rng, map_bp, map_cm)
    with Haps(haps, sample, gmap, 5.0) as h:  # the reference's default --memory
        assert (h.N, h.L, h.num_chunks) == (N, L, 1)
        first, lc, wb = h.chunk(0)
        assert (first, lc) == (0, L) and wb[0] == 0 and wb[-1] == L
        _, _, rpos, r = h.positions()
        assert np.array_equal(r, np.maximum(np.diff(rpos), 1e-10) * 2500)  # src/data.cpp:484-488
        plan = ra.Plan(0)
        h.load_chunk(plan, 0)
    W, R = len(wb) - 1, ra.record_bytes(N)
    out = torch.empty(W * N * R, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    plan.prepare(0.001, 0.001, 0, N, st)
    plan.paint(out.data_ptr(), st)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(W, N, R)
    plan.close()
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb)
    assert compare_records(got, want, N, "config 1 (200 haplotypes, 5 Mb) from .haps") == 0.0


def test_rows_of_the_wrong_shape_are_refused():
    import relate_amd as ra
    p = ra.Plan(0)
    with pytest.raises(ValueError):
        p.load_rows_host(130, np.zeros((50, 3), dtype=np.uint64), np.ones(50), [0, 50])
    p.close()
