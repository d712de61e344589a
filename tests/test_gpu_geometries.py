"""Kernel geometries that the other GPU tests do not reach (run with -m gpu): 36 to 80 states per lane,
i.e. large panels painted with fewer waves per target than the automatic choice, one theta and two.
Targets are sampled (a range of 64 and a range of 30) so that the CPU oracle stays quick."""
import pytest

import oracle_lib as ol
import synth_np as sy
from test_gpu_parity import _ra, compare_records, gpu_paint

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,L,W,seed,waves", [
    (5008, 120, 2, 31, 2),   # 2 x 40 words
    (5008, 120, 2, 31, 1),   # 1 x 80 words
    (5008, 120, 2, 31, 8),   # falls back to 4 x 20: 8 x 12 words would run past the padded row
    (3000, 150, 2, 32, 1),   # 1 x 48
    (3000, 150, 2, 32, 2),   # 2 x 24
    (4500, 100, 2, 33, 4),   # 4 x 20
    (2300, 200, 3, 34, 1),   # 1 x 36
    (10000, 60, 2, 35, 0),   # the automatic choice for a panel of 10 000: 4 x 40
    (5600, 100, 2, 38, 0),   # 88 words of bits: no geometry is 88 words wide, the rows are padded to 96 (4 x 24)
    (5600, 100, 2, 38, 1),   # one wave cannot hold 88 words: 2 x 48
    (6500, 80, 2, 39, 0),    # 102 words, rows padded to 112 (4 x 28)
    (6500, 80, 2, 39, 8),    # 8 x 16 words would run past the padded row: 4 x 28
])
def test_geometry(N, L, W, seed, waves):
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=seed, r_mode="skew" if seed % 2 else "flat")
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb, 0, 64)
    got = gpu_paint(seq, r, 0.001, 0.001, wb, 1, waves, 0, 64)
    assert compare_records(got, want, N, f"N={N} waves={waves}") == 0.0
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.01, wb, 100, 130)
    got = gpu_paint(seq, r, 0.001, 0.01, wb, 1, waves, 100, 130)
    assert compare_records(got, want, N, f"N={N} waves={waves}, two thetas") == 0.0


def test_automatic_choice_counts_rounds():
    """With all 3000 targets in one launch the automatic choice is one wave of 48 words per target (two
    rounds of waves over the chip) rather than the 2 x 24 of the size rule (three rounds); with 64 targets the
    chip is not full and more waves per target win.  Every record of the full launch against the oracle."""
    N, L, W = 3000, 100, 2
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=36, r_mode="flat")
    ra = _ra()
    plan = ra.Plan(0, 1, 0)
    plan.load_host(seq, r, wb)
    plan.prepare(0.001, 0.001, 0, N)
    assert plan.geometry() == {"waves_per_target": 1, "states_per_lane": 48}
    plan.prepare(0.001, 0.001, 0, 64)
    assert plan.geometry()["waves_per_target"] >= 2
    plan.close()
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb, 0, N)
    got = gpu_paint(seq, r, 0.001, 0.001, wb, 1, 0, 0, N)
    assert compare_records(got, want, N, "N=3000, all targets, automatic geometry") == 0.0


def test_trace_of_a_launch(tmp_path, monkeypatch):
    """RELATE_TRACE=<file>: per workgroup of the recursion kernel 4 x u64 -- where it ran, start and end in 100 MHz
    ticks, pass | list length << 8.  The start is written when the workgroup begins and the rest when it ends;
    both passes of every target are there, each with the target's number of sites."""
    import numpy as np
    N, L, W = 300, 400, 3
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=37, r_mode="flat")
    path = tmp_path / "trace.bin"
    monkeypatch.setenv("RELATE_TRACE", str(path))
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb, 0, N)
    got = gpu_paint(seq, r, 0.001, 0.001, wb, 1, 0, 0, N)
    assert compare_records(got, want, N, "traced launch") == 0.0
    tr = np.fromfile(path, dtype=np.uint64).reshape(2 * N, 4)
    assert np.all(tr[:, 1] > 0) and np.all(tr[:, 2] >= tr[:, 1]), "start and end times of every workgroup"
    passes = (tr[:, 3] & np.uint64(0xFF)).astype(np.int64)
    nd = (tr[:, 3] >> np.uint64(8)).astype(np.int64)
    assert sorted(passes.tolist()) == [0] * N + [1] * N
    der = seq == ord("1")
    want_nd = der.sum(axis=0) + (~der[0]).astype(np.int64) + (~der[-1]).astype(np.int64)  # sites 0 and L-1 are always listed
    assert sorted(nd[passes == 0].tolist()) == sorted(want_nd.tolist()) == sorted(nd[passes == 1].tolist())
