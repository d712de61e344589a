"""GPU parity of the tree builder's opening scans (MinMatch::Initialize, src/tree_builder.cpp:57-127) through the C
ABI: relate_plan_minmatch_scan on matrices in HBM -- synthetic ones with many ties, and the matrices
relate_distance_get_matrix leaves there for SNPs of a painted chunk -- against the restated oracle: min_values bit
for bit, the pair list entry for entry in the reference's loop order; then the host half with the reference
generator's draws against the oracle's Initialize.  (The file name makes pytest collect it late on purpose: a run with -x stops at
the first failure, and one here, in the newest entry point, must not hide the tests of the hot path.)"""
import numpy as np
import pytest

import oracle_lib as ol  # noqa: F401
import synth_np as sy
import treebuilder_lib as tb

pytestmark = pytest.mark.gpu


def _same(got, want):
    assert np.array_equal(got.min_values.view(np.uint32), want[0].view(np.uint32)), "min_values"
    assert got.count == len(want[1])
    assert np.array_equal(got.cand_i, want[1]) and np.array_equal(got.cand_j, want[2])
    assert np.array_equal(got.cand_dist.view(np.uint32), want[3].view(np.uint32))


@pytest.mark.parametrize("N,seed,ties", [(5, 0, True), (64, 1, True), (130, 3, False), (1000, 4, True), (5008, 5, True)])
def test_scan_of_synthetic_matrices(N, seed, ties):
    import torch
    import relate_amd as ra
    d = tb.distance_like_matrix(N, seed, ties)
    thr = ra.minmatch_threshold(0.001)
    plan = ra.Plan(0)
    dm = torch.from_numpy(d).cuda()
    torch.cuda.synchronize()
    got = ra.MinMatchScan(plan, N, dm.data_ptr(), thr)
    _same(got, tb.oracle_scan(d, thr))
    plan.close()


def test_every_pair_a_candidate_grows_the_arrays():
    import torch
    import relate_amd as ra
    N = 300  # 44 850 pairs: more than the first guess of the mirror's arrays
    dm = torch.zeros((N, N), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    plan = ra.Plan(0)
    got = ra.MinMatchScan(plan, N, dm.data_ptr(), 0.5)
    assert got.count == N * (N - 1) // 2
    _same(got, tb.oracle_scan(np.zeros((N, N), dtype=np.float32), 0.5))
    plan.close()


def test_scan_of_the_distance_matrices_of_a_painted_chunk():
    """The matrices the tree builder really sees: GetMatrix at the first SNP of a section and at SNPs inside it."""
    import torch
    import relate_amd as ra
    N, L, W, theta = 200, 400, 2, 0.001
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=91, r_mode="flat")
    rpos = np.concatenate([[0.0], np.cumsum(r / 2500.0)])
    plan = ra.Plan(0)
    plan.load_host(seq, r, wb)
    plan.prepare(theta, theta, 0, N)
    stones = torch.zeros(plan.out_bytes(), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    plan.paint(stones.data_ptr())
    plan.timings()
    dmeas = ra.NativeDistanceMeasure(plan, seq, rpos, stones)
    thr = ra.minmatch_threshold(theta)
    total = 0
    for snp in range(0, 60):
        if snp > 0:
            dmeas.advance(snp)
        if snp not in (0, 1, 17, 59):
            continue
        m = dmeas.GetMatrix(snp)
        torch.cuda.synchronize()
        got = ra.MinMatchScan(plan, N, m.data_ptr(), thr)
        want = tb.oracle_scan(m.cpu().numpy(), thr)
        _same(got, want)
        assert np.all(got.min_values == np.float32(thr))  # GetMatrix subtracts every row's minimum (:194-196)
        total += got.count
        # the host half: the candidates and the best pair of the oracle's Initialize on the same matrix
        (l1, l2, ds, d2), best, _ = tb.initialize(m.cpu().numpy(), theta)
        draws = iter(tb.uniform_draws(got.count))
        mc, b = ra.apply_candidates(N, got.cand_i, got.cand_j, got.cand_dist, lambda: next(draws))
        assert [c.lin1 for c in mc] == list(l1) and [c.lin2 for c in mc] == list(l2)
        assert (b.lin1, b.lin2, b.dist, b.dist2) == best[:4]
    assert total > 0
    dmeas.close()
    plan.close()
