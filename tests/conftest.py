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


GOLDEN_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    return load_golden(request.param)
