"""GPU parity of the distance matrix (DistanceMeasure::GetMatrix, src/anc_builder.cpp:113-211) on a topology
that stays in HBM: the window is re-painted on the GPU, the per-row bookkeeping of the reference (which
topology row precedes the SNP, whether the row is taken as is or mixed with the next, the two weights) is
restated here on the host, and the N x N matrix of the GPU kernel is compared with the CPU oracle fed with
the same (downloaded) topology.

The oracle of this function is a restatement without a compiled reference behind it (parity unpinned, see
oracle/distance_oracle.c).  Bar: every entry bit-exact, direct and mixed rows alike: the kernel's operations are
individually rounded IEEE operations in the reference's order, and the one libm call per mixed row (exp, on the
difference of two logscales) is made on the host with the C library, as the reference makes it."""
import numpy as np
import pytest

import oracle_lib as ol
import synth_np as sy
from test_gpu_repaint import _ra

pytestmark = pytest.mark.gpu


def bookkeeping(seq, rpos, wb, w, snp, a_snp, off):
    """v_snp_prev / interpolation flags / weights of src/anc_builder.cpp:85-105,:122,:144-161 for one SNP of
    window w.  a_snp[n] = site of target n's alpha stone (row 0 of its topology)."""
    L, N = seq.shape
    der = seq == ord("1")
    row = np.zeros(N, dtype=np.int64)
    interp = np.zeros(N, dtype=np.uint8)
    wl = np.zeros(N)
    wr = np.zeros(N)
    for n in range(N):
        # rows of target n: its alpha-stone site, then every derived site after it
        v = int(der[a_snp[n] + 1:snp + 1, n].sum())
        row[n] = off[n] + v
        if der[snp, n] or snp == 0 or snp == L - 1:
            continue
        interp[n] = 1
        t = snp
        while not der[t, n] and t > 0:
            t -= 1
        rp = rpos[t]
        nxt = snp
        while not (der[nxt, n] or nxt == L - 1):
            nxt += 1
        rn = rpos[nxt]
        if rp == rn:
            wl[n] = wr[n] = 0.5
        else:
            wl[n] = (rn - rpos[snp]) / (rn - rp)
            wr[n] = (rpos[snp] - rp) / (rn - rp)
    return row, interp, wl, wr


@pytest.mark.parametrize("N,L,W,seed,theta", [(70, 1200, 3, 51, 0.001), (200, 900, 2, 52, 0.01), (513, 600, 2, 53, 0.001)])
def test_distance_matrix_matches_oracle(N, L, W, seed, theta):
    import torch
    ra = _ra()
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=seed, r_mode="flat")
    rpos = np.concatenate([[0.0], np.cumsum(r / 2500.0)])  # genetic positions consistent with r (src/data.cpp:484-488)
    plan = ra.Plan(0, 1, 0)
    plan.load_host(seq, r, wb)
    plan.prepare(theta, theta, 0, N)
    stones = torch.zeros(plan.out_bytes(), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    plan.paint(stones.data_ptr())
    plan.timings()
    R = ra.record_bytes(N)
    for w in range(W):
        rows, pitch, off = plan.repaint_layout(w)
        top = torch.zeros((rows, pitch), dtype=torch.float32, device="cuda:0")
        ls = torch.zeros((rows,), dtype=torch.float32, device="cuda:0")
        mat = torch.zeros((N, N), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        plan.repaint(w, stones.data_ptr() + w * N * R, top.data_ptr(), ls.data_ptr())
        d = ol.decode_records(stones.cpu().numpy().reshape(plan.W, N, R)[w:w + 1])
        a_snp = d["a_snp"][0]
        h_top, h_ls = top.cpu().numpy(), ls.cpu().numpy()
        lo, hi = int(wb[w]), int(wb[w + 1]) - 1
        for snp in sorted({lo, (lo + hi) // 2, hi, min(hi, lo + 7)}):
            row, interp, wl, wr = bookkeeping(seq, rpos, wb, w, snp, a_snp, off)
            assert np.all(row + interp < off[1:]), "bookkeeping left the section"
            plan.distance_matrix(N, top.data_ptr(), pitch, ls.data_ptr(), rows, row, interp, wl, wr, mat.data_ptr())
            got = mat.cpu().numpy()
            want = ol.distance_matrix(h_top, h_ls, row, interp, wl, wr, N)
            assert np.isfinite(got).all()
            assert (np.diag(got) == 0).all()
            assert interp.any() or snp in (0, L - 1)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"w{w} snp{snp}: {np.count_nonzero(got != want)} entries differ"
            offdiag = ~np.eye(N, dtype=bool)
            assert np.all(got[offdiag].reshape(N, N - 1).min(axis=1) <= 1e-4), "row minimum was not subtracted"
    plan.close()


def test_distance_measure_mirror_walks_a_chunk():
    """relate_amd.DistanceMeasure driven the way AncesTreeBuilder::BuildTopology drives the reference's
    (src/anc_builder.cpp:279-370): GetMatrix at the first SNP of a section, then advance + GetMatrix SNP by SNP,
    across a section change.  Its incremental bookkeeping must agree with the closed form above, and every
    matrix with the oracle."""
    import torch
    ra = _ra()
    N, L, W, theta = 60, 260, 2, 0.001
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=61, r_mode="flat")
    rpos = np.concatenate([[0.0], np.cumsum(r / 2500.0)])
    plan = ra.Plan(0, 1, 0)
    plan.load_host(seq, r, wb)
    plan.prepare(theta, theta, 0, N)
    stones = torch.zeros(plan.out_bytes(), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    plan.paint(stones.data_ptr())
    plan.timings()
    R = ra.record_bytes(N)
    host = stones.cpu().numpy().reshape(W, N, R)
    d = ra.DistanceMeasure(plan, seq, rpos, stones)
    checked = 0
    for w in range(W):
        lo, hi = int(wb[w]), int(wb[w + 1]) - 1
        a_snp = ol.decode_records(host[w:w + 1])["a_snp"][0]
        for snp in range(lo, hi + 1):
            if snp > lo:
                d.advance(snp)
            got = d.GetMatrix(snp).cpu().numpy()
            assert (d.section_startpos, d.section_endpos) == (lo, hi)
            row, interp, wl, wr = bookkeeping(seq, rpos, wb, w, snp, a_snp, d.row_offsets)
            g_row, g_interp, g_wl, g_wr = d._last_args
            assert np.array_equal(g_row, row) and np.array_equal(g_interp, interp), f"snp {snp}"
            assert np.allclose(g_wl, wl, rtol=1e-12, atol=0) and np.allclose(g_wr, wr, rtol=1e-12, atol=0)
            if snp in (lo, lo + 1, (lo + hi) // 2, hi):
                want = ol.distance_matrix(d.topology.cpu().numpy(), d.logscales.cpu().numpy(), row, interp, wl, wr, N)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
                checked += 1
    assert checked == 4 * W
    plan.close()


def test_library_distance_measure_equals_the_python_mirror():
    """relate_distance_* (the DistanceMeasure a C++ caller links against) walked over a chunk beside
    relate_amd.DistanceMeasure: the same bookkeeping at every SNP and bit-identical matrices (same kernels, same
    arguments), across a section change and with an explicit GetTopologyWithRepaint in the middle of a section."""
    import torch
    ra = _ra()
    N, L, W, theta = 75, 300, 3, 0.001
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=71, r_mode="flat")
    rpos = np.concatenate([[0.0], np.cumsum(r / 2500.0)])
    plan = ra.Plan(0, 1, 0)
    plan.load_host(seq, r, wb)
    plan.prepare(theta, theta, 0, N)
    stones = torch.zeros(plan.out_bytes(), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    plan.paint(stones.data_ptr())
    plan.timings()
    a = ra.DistanceMeasure(plan, seq, rpos, stones)
    b = ra.NativeDistanceMeasure(plan, seq, rpos, stones)
    for snp in range(0, L):
        if snp > 0:
            a.advance(snp)
            b.advance(snp)
        ma = a.GetMatrix(snp).cpu().numpy().copy()
        torch.cuda.synchronize()
        mb = b.GetMatrix(snp).cpu().numpy()
        st = b.state()
        assert (st["section"], st["section_startpos"], st["section_endpos"]) == (a.section, a.section_startpos, a.section_endpos)
        assert np.array_equal(st["v_snp_prev"], a.v_snp_prev), f"snp {snp}"
        assert np.array_equal(st["v_rpos_prev"], a.v_rpos_prev) and np.array_equal(st["v_rpos_next"], a.v_rpos_next)
        assert np.array_equal(ma.view(np.uint32), mb.view(np.uint32)), f"snp {snp}"
    assert a.section == W
    # entering a chunk in the middle of a section (the reference's Assign / GetTopologyWithRepaint with snp > start)
    mid = (int(wb[1]) + int(wb[2])) // 2
    c = ra.NativeDistanceMeasure(plan, seq, rpos, stones, section=1)
    c.GetTopologyWithRepaint(mid)
    e = ra.DistanceMeasure(plan, seq, rpos, stones, section=1)
    e.GetTopologyWithRepaint(mid)
    st = c.state()
    assert np.array_equal(st["v_snp_prev"], e.v_snp_prev) and np.array_equal(st["v_rpos_prev"], e.v_rpos_prev)
    assert np.array_equal(c.GetMatrix(mid).cpu().numpy().view(np.uint32), e.GetMatrix(mid).cpu().numpy().view(np.uint32))
    with pytest.raises(ra.RelateError):  # a SNP outside the next section
        ra.NativeDistanceMeasure(plan, seq, rpos, stones, section=1).GetTopologyWithRepaint(int(wb[1]) - 1)
    for o in (b, c):
        o.close()
    plan.close()
