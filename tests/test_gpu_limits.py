"""Limits and misuse of the GPU entry points: sizes at and past what the kernels support, and calls out of
order, must end in an error code (never in a fault)."""
import numpy as np
import pytest

import oracle_lib as ol
import synth_np as sy
from test_gpu_parity import compare_records, gpu_paint

pytestmark = pytest.mark.gpu


def _ra():
    import relate_amd as ra
    return ra


def random_chunk(N, L, seed):
    rng = np.random.default_rng(seed)
    seq = np.where(rng.random((L, N)) < 0.3, ord("1"), ord("0")).astype(np.uint8)
    r = rng.uniform(0.001, 0.05, L)
    return seq, r


@pytest.mark.parametrize("N", [20000, 36864])
def test_widest_panels(N):
    """Eight waves per target, more than 64 KiB of LDS for the stone staging."""
    ra = _ra()
    L = 6
    seq, r = random_chunk(N, L, N)
    wb = np.array([0, 3, L], dtype=np.int32)
    ks = [0, 1, 63, 64, N // 2, N - 1]
    for k in ks:
        got = gpu_paint(seq, r, 0.001, 0.001, wb, 1, 0, k, k + 1)
        want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb, k, k + 1)
        compare_records(got, want, N, f"N={N} k={k}")


def test_panel_too_wide_is_refused():
    ra = _ra()
    N, L = 36865, 4
    seq, r = random_chunk(N, L, 1)
    wb = np.array([0, L], dtype=np.int32)
    with pytest.raises(ra.RelateError) as e:
        ra.paint_stones(seq, r, 0.001, 0.001, wb, 0, 1)
    assert e.value.code == -5  # RELATE_ERANGE


def test_misuse_is_refused():
    import torch
    ra = _ra()
    seq, r, wb = sy.make_chunk(N=40, L=300, W=2, seed=3, r_mode="flat")
    with pytest.raises(ra.RelateError):
        ra.paint_stones(seq, r, 0.001, 0.001, np.array([0, 100, 299], dtype=np.int32))   # last boundary != L
    with pytest.raises(ra.RelateError):
        ra.paint_stones(seq, r, 0.001, 0.001, np.array([0, 200, 100, 300], dtype=np.int32))  # decreasing
    with pytest.raises(ra.RelateError):
        ra.paint_stones(seq, r, 0.001, 0.001, wb, 5, 41)    # target range past N
    with pytest.raises(ra.RelateError):
        ra.paint_stones(seq, r, 1.5, 0.001, wb)             # theta out of (0, 1)
    with pytest.raises(ra.RelateError):
        ra.Plan(0, 7, 0)                                    # unknown kernel variant
    with pytest.raises(ra.RelateError):
        ra.Plan(0, 1, 3)                                    # waves per target must be 0, 1, 2, 4 or 8
    plan = ra.Plan(0, 1, 0)
    out = torch.zeros(1024, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(ra.RelateError):
        plan.paint(out.data_ptr())                          # nothing loaded or prepared
    plan.load_host(seq, r, wb)
    with pytest.raises(ra.RelateError):
        plan.paint(out.data_ptr())                          # not prepared
    with pytest.raises(ra.RelateError):
        plan.repaint_layout(0)                              # not prepared
    plan.prepare(0.001, 0.001, 0, 40)
    with pytest.raises(ra.RelateError):
        plan.repaint_layout(2)                              # window out of range
    plan.close()
    plan0 = ra.Plan(0, 0, 0)                                # the plain cross-check kernel has no re-painting
    plan0.load_host(seq, r, wb)
    plan0.prepare(0.001, 0.001, 0, 40)
    with pytest.raises(ra.RelateError):
        plan0.repaint_layout(0)
    plan0.close()
