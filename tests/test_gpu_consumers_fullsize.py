"""The two consumers of the stepping stones at the benchmark's sizes on the GPU: one window of the 2000-haplotype
chunk (N=2000, L=50 000) and one of the 5008-haplotype chunk (N=5008, L=100 000), painted there, re-painted there
(FastPainting::RePaintSection for all targets, src/fast_painting.cpp:654-1126) and turned into distance matrices
(DistanceMeasure::GetMatrix, src/anc_builder.cpp:113-211).

Sampled targets of the re-painted window against oracle_repaint_section (pinned: byte-identical to the compiled
reference), with the share of floats that are not bit-identical recorded (gpurun_out/fullsize_bit_identity.txt) and
bounded; sampled SNPs' whole N x N matrices against oracle_distance_matrix (a restatement, parity unpinned) bit for
bit."""
import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_distance import bookkeeping
from test_gpu_fullsize import _note
from test_gpu_repaint import check_window

pytestmark = pytest.mark.gpu


def bookkeeping_fast(seq, rpos, snp, a_snp, off):
    """test_gpu_distance.bookkeeping with the per-haplotype scans vectorised (N in the thousands)."""
    L, N = seq.shape
    der = seq == ord("1")
    row = np.zeros(N, dtype=np.int64)
    interp = np.zeros(N, dtype=np.uint8)
    wl = np.zeros(N)
    wr = np.zeros(N)
    lo = int(a_snp.min())
    csum = np.cumsum(der[lo:snp + 1], axis=0, dtype=np.int64)  # carriers of rows lo..snp
    for n in range(N):
        a = int(a_snp[n])
        v = int(csum[snp - lo, n] - csum[a - lo, n])
        row[n] = off[n] + v
        if der[snp, n] or snp == 0 or snp == L - 1:
            continue
        interp[n] = 1
        col = der[:, n]
        prev = np.flatnonzero(col[:snp + 1])
        t = int(prev[-1]) if len(prev) else 0
        nxt_c = np.flatnonzero(col[snp:])
        nxt = snp + int(nxt_c[0]) if len(nxt_c) else L - 1
        rp, rn = rpos[t], rpos[nxt]
        if rp == rn:
            wl[n] = wr[n] = 0.5
        else:
            wl[n] = (rn - rpos[snp]) / (rn - rp)
            wr[n] = (rpos[snp] - rp) / (rn - rp)
    return row, interp, wl, wr


@pytest.mark.timeout(900)
@pytest.mark.parametrize("N,L,tag,waves", [(2000, 50000, "c2", 0), (5008, 100000, "c3", 0),
                                           (2000, 50000, "c2, one wave of 32 words per target: every alpha row a checkpoint", 1)])
def test_one_window_repainted_and_measured(N, L, tag, waves):
    import torch
    import relate_amd as ra
    theta = 0.001
    seq, r = ra.synth_chunk(N, L, 2)  # the chunks bench.py times
    wb = ra.window_boundaries(seq, memory_gb=5.0)
    W, R = len(wb) - 1, ra.record_bytes(N)
    rpos = np.concatenate([[0.0], np.cumsum(r / 2500.0)])
    d_seq = torch.from_numpy(seq).cuda()
    d_r = torch.from_numpy(r).cuda()
    stones = torch.empty(W * N * R, dtype=torch.uint8, device="cuda")
    plan = ra.Plan(0, 1, waves)
    st = torch.cuda.current_stream().cuda_stream
    plan.load_device(N, L, d_seq.data_ptr(), d_r.data_ptr(), wb, st)
    plan.prepare(theta, theta, 0, N, st)
    plan.paint(stones.data_ptr(), st)
    torch.cuda.synchronize()  # (d_r stays alive: the plan borrows it, and re-painting reads it again)
    w = W // 2
    if waves == 1:
        assert plan.repaint_geometry()["checkpoint_rows"] == 1
    rows, pitch, off = plan.repaint_layout(w)
    top = torch.full((rows, pitch), float("nan"), dtype=torch.float32, device="cuda")
    ls = torch.full((rows,), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    plan.repaint(w, stones.data_ptr() + w * N * R, top.data_ptr(), ls.data_ptr(), st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(top).all()) and bool(torch.isfinite(ls).all()) and bool((top >= 0).all())
    window_stones = stones.view(W, N, R)[w].cpu().numpy().reshape(1, N, R)
    h_top, h_ls = top.cpu().numpy(), ls.cpu().numpy()
    # re-painting: sampled targets (first, last, around word and wave edges, random) against the pinned oracle
    rng = np.random.default_rng(7)
    picks = sorted(set([0, 1, 63, 64, N // 2 - 1, N // 2, N - 2, N - 1] + [int(v) for v in rng.integers(0, N, 4)]))
    frac = check_window(seq, r, theta, window_stones, 0, h_top, h_ls, off, pitch, picks)  # (its window index counts inside the array handed over)
    _note(f"{tag} N={N} L={L} window {w} of {W}: {rows} topology rows re-painted, {len(picks)} targets against the oracle, "
          f"share of floats not bit-identical {frac:.3e}")
    assert frac == 0.0, f"{frac:.2e} of the re-painted floats differ from the oracle's in the last bit (measured: none)"
    # distance matrices: first, middle and last SNP of the window, all N x N entries
    a_snp = ol.decode_records(window_stones)["a_snp"][0]
    lo, hi = int(wb[w]), int(wb[w + 1]) - 1
    mat = torch.zeros((N, N), dtype=torch.float32, device="cuda")
    for snp in sorted({lo, (lo + hi) // 2, hi}):
        row, interp, wl, wr = bookkeeping_fast(seq, rpos, snp, a_snp, off)
        if N <= 2000 and snp == lo:  # the vectorised bookkeeping against the plain one
            row2, interp2, wl2, wr2 = bookkeeping(seq, rpos, wb, w, snp, a_snp, off)
            assert np.array_equal(row, row2) and np.array_equal(interp, interp2) and np.array_equal(wl, wl2) and np.array_equal(wr, wr2)
        assert np.all(row + interp < off[1:]), "bookkeeping left the section"
        plan.distance_matrix(N, top.data_ptr(), pitch, ls.data_ptr(), rows, row, interp, wl, wr, mat.data_ptr(), st)
        torch.cuda.synchronize()
        got = mat.cpu().numpy()
        want = ol.distance_matrix(h_top, h_ls, row, interp, wl, wr, N)
        assert np.isfinite(got).all() and (np.diag(got) == 0).all()
        nbad = int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
        _note(f"{tag} distance matrix at SNP {snp}: {int(interp.sum())} of {N} rows mixed, {nbad} of {N * N} entries not bit-identical")
        assert nbad == 0
    plan.close()
