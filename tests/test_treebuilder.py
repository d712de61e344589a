"""The opening scans of the reference's tree builder (MinMatch::Initialize, src/tree_builder.cpp:57-127): the
restated oracle against the reference's own known answers (test/test_treebuilder.cpp:36-86,:88-137), the library's
scan kernels -- run here on CPU threads from the very source the GPU build compiles -- against the oracle, and the
host half (one draw per candidate pair, relate_minmatch_apply_candidates) against the oracle's Initialize."""
import numpy as np
import pytest

import relate_amd as ra
import treebuilder_lib as tb


def test_threshold_matches():
    for theta in (0.025, 0.001, 0.01):
        assert ra.minmatch_threshold(theta) == tb.threshold(theta) == float(np.float32(-0.2 * np.log(theta / (1 - theta))))


@pytest.mark.parametrize("case", tb.fixtures()["cases"], ids=lambda c: c["name"])
def test_oracle_reproduces_the_reference_known_answers(case):
    theta = tb.fixtures()["theta"]
    d = np.array(case["d"], dtype=np.float32)
    N = d.shape[0]
    thr = tb.threshold(theta)
    mv, ci, cj, cd = tb.oracle_scan(d, thr)
    assert np.array_equal(mv, np.array(case["row_min"], dtype=np.float32) + np.float32(thr))
    assert [[int(a), int(b), float(c)] for a, b, c in zip(ci, cj, cd)] == case["candidates"]
    for ages in (None, np.zeros(N)):  # the reference's test passes sample ages of zero (:31); without them :57-127 runs
        parents, fallbacks = tb.quick_build(d, theta, ages)
        assert list(parents) == case["parents"], f"ages={ages is not None}"
        assert fallbacks == case["sym_fallbacks"]


def test_oracle_builds_a_tree_from_an_empty_matrix():
    """test/test_treebuilder.cpp:24-32: QuickBuild on the all-zero matrix must terminate with a binary tree."""
    parents, fallbacks = tb.quick_build(np.zeros((5, 5), dtype=np.float32), 0.025, np.zeros(5))
    assert fallbacks == 0 and sorted(np.bincount(parents[:-1])[5:]) == [2, 2, 2, 2] and parents[-1] == -1


CASES = [(5, 0, True), (64, 1, True), (65, 2, True), (130, 3, False), (200, 4, True), (333, 5, True)]


@pytest.mark.parametrize("N,seed,ties", CASES)
def test_scan_kernels_on_cpu_threads_match_the_oracle(N, seed, ties):
    d = tb.distance_like_matrix(N, seed, ties)
    thr = tb.threshold(0.001)
    want = tb.oracle_scan(d, thr)
    got = tb.hostrun_scan(d, thr)
    assert len(want[1]) > 0
    for g, w, name in zip(got, want, ("min_values", "cand_i", "cand_j", "cand_dist")):
        assert np.array_equal(g.view(np.uint32) if g.dtype == np.float32 else g,
                              w.view(np.uint32) if w.dtype == np.float32 else w), name


def test_scan_kernels_on_the_reference_matrices_and_edge_values():
    theta = tb.fixtures()["theta"]
    thr = tb.threshold(theta)
    for case in tb.fixtures()["cases"]:
        d = np.array(case["d"], dtype=np.float32)
        got = tb.hostrun_scan(d, thr)
        assert [[int(a), int(b), float(c)] for a, b, c in zip(*got[1:])] == case["candidates"]
    # all entries equal: every pair is a candidate; a NaN never becomes a minimum nor passes a test; infinities
    d = np.zeros((70, 70), dtype=np.float32)
    got, want = tb.hostrun_scan(d, thr), tb.oracle_scan(d, thr)
    assert len(got[1]) == 70 * 69 // 2 and all(np.array_equal(g, w) for g, w in zip(got, want))
    d = tb.distance_like_matrix(70, 9)
    d[3, 5] = np.nan
    d[7, :] = np.inf
    d[7, 7] = 0
    d[9, 11] = -0.0
    got, want = tb.hostrun_scan(d, thr), tb.oracle_scan(d, thr)
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:]))


@pytest.mark.parametrize("with_ages", [False, True])
def test_host_half_of_initialize_matches_the_oracle(with_ages):
    """relate_minmatch_apply_candidates fed with the oracle's pair list and the reference generator's draws."""
    N, theta, Ne = 120, 0.001, 30000
    d = tb.distance_like_matrix(N, 21)
    ages = None
    age = 0.0
    if with_ages:
        ages = np.zeros(N)
        ages[::7] = 50.0   # some older samples
        ages[::11] = 4000.0
        n0 = int((ages == ages.min()).sum())
        age = ages.min() + 2.0 / (n0 * (n0 - 1.0)) * Ne
    (l1, l2, ds, d2), best, mv = tb.initialize(d, theta, ages, Ne)
    _, ci, cj, cd = tb.oracle_scan(d, tb.threshold(theta))
    draws = iter(tb.uniform_draws(len(ci)))
    m, b = ra.apply_candidates(N, ci, cj, cd, lambda: next(draws), ages, age)
    assert [c.lin1 for c in m] == list(l1) and [c.lin2 for c in m] == list(l2)
    assert np.array_equal(np.array([c.dist for c in m]), ds) and np.array_equal(np.array([c.dist2 for c in m]), d2)
    assert (b.lin1, b.lin2, b.dist, b.dist2) == best[:4]
    assert len(set(zip(l1, l2))) > 3
