import glob
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_usable():
    """A GPU test can run: the library has been built and sees a device."""
    try:
        import relate_amd as ra
        return ra.lib().relate_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """Without a device (or without the built library) the tests marked `gpu` are skipped, not failed, so that a
    plain `pytest` works on a CPU-only machine; `-m gpu` on a GPU box runs them."""
    if not any("gpu" in item.keywords for item in items):
        return
    if _gpu_usable():
        return
    skip = pytest.mark.skip(reason="needs an MI355X and the built librelate_gpu.so")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_fast_log_golden():
    """(x, y): float32 arguments and the reference's fast_log of them, bit for bit (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(HERE, "golden", "fast_log.npz"), allow_pickle=False)
    return z["x_bits"].view(np.float32), z["y_bits"].view(np.float32)


def load_golden(name):
    """Golden fixture -> dict with seq unpacked to the (L,N) '0'/'1' byte matrix."""
    z = np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    N, L = int(d["N"]), int(d["L"])
    bits = np.unpackbits(d["seq_bits"])[: L * N].reshape(L, N)
    d["seq"] = (bits + ord("0")).astype(np.uint8)
    d["N"], d["L"] = N, L
    d["theta_painter"] = float(d["theta_painter"])
    d["theta_data"] = float(d["theta_data"])
    return d


# painting fixtures (inputs + the reference's records); fast_log.npz is a fixture of another kind
GOLDEN_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz"))
                      if os.path.basename(p) != "fast_log.npz")


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    return load_golden(request.param)


def pytest_sessionfinish(session, exitstatus):
    """Every GPU-against-oracle comparison of the session with its count of floats that were not bit-identical,
    written where a GPU run's outputs are collected (gpurun_out/): the measurement behind the == 0 assertions."""
    mod = sys.modules.get("test_gpu_parity")
    log = getattr(mod, "BIT_LOG", None)
    if not log:
        return
    try:
        d = os.path.join(os.path.dirname(HERE), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bit_identity_tests.txt"), "w") as f:
            tot = sum(t for _, t, _ in log)
            bad = sum(b for _, _, b in log)
            f.write(f"# {len(log)} comparisons, {tot} floats, {bad} not bit-identical\n")
            for tag, t, b in log:
                f.write(f"{b:8d} of {t:10d}  {tag}\n")
    except OSError:
        pass
