"""GPU parity of the re-painting path (FastPainting::RePaintSection, src/fast_painting.cpp:654-1126) through
the C ABI: the stones of a window, painted on the GPU, are re-painted on the GPU for all targets at once and
compared, target by target, with the CPU oracle fed with the same stones.

Bar: row counts and zero pattern exact; floats (topology, logscales) within 1e-5 relative, and the
log-probability log(topology) + logscale, which is what DistanceMeasure::GetMatrix consumes
(src/anc_builder.cpp:122-141), within 1e-5 of its magnitude.  Asserted on top of that, because it is what was
measured: no topology float differs from the oracle's in any bit."""
import numpy as np
import pytest

import oracle_lib as ol
import synth_np as sy

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5
# Share of re-painted floats whose last bit may differ from the oracle's.  The bar is REL_TOL; what is asserted is what
# was measured on every case here and on the benchmark's windows (gpurun_out/repaint_bit_identity.txt): none.
MAX_SHARE_NOT_IDENTICAL = 0.0


def _record(line):
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "repaint_bit_identity.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _ra():
    import relate_amd as ra
    return ra


def gpu_paint_and_repaint(seq, r, theta, wb, windows, waves=0):
    import torch
    ra = _ra()
    L, N = seq.shape
    plan = ra.Plan(0, 1, waves)
    plan.load_host(seq, r, wb)
    plan.prepare(theta, theta, 0, N)
    stones = torch.zeros(plan.out_bytes(), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    plan.paint(stones.data_ptr())
    plan.timings()
    R = ra.record_bytes(N)
    host_stones = stones.cpu().numpy().reshape(plan.W, N, R)
    res = {}
    for w in windows:
        rows, pitch, off = plan.repaint_layout(w)
        top = torch.full((rows, pitch), float("nan"), dtype=torch.float32, device="cuda:0")
        ls = torch.full((rows,), float("nan"), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        plan.repaint(w, stones.data_ptr() + w * N * R, top.data_ptr(), ls.data_ptr())
        plan.timings()
        res[w] = (top.cpu().numpy(), ls.cpu().numpy(), off, pitch)
    plan.close()
    return host_stones, res


def check_window(seq, r, theta, host_stones, w, top, ls, off, pitch, targets=None):
    L, N = seq.shape
    d = ol.decode_records(host_stones[w:w + 1])
    assert pitch >= N and pitch % 64 == 0
    nbad = 0
    total = 0
    for k in (range(N) if targets is None else targets):
        first, last = int(d["a_snp"][0, k]), int(d["b_snp"][0, k])
        want_top, want_ls = ol.oracle().repaint_section(seq, r, theta, first, last, d["alpha"][0, k], d["beta"][0, k],
                                                        float(d["a_ls"][0, k]), float(d["b_ls"][0, k]), k)
        nrow = off[k + 1] - off[k]
        assert nrow == len(want_top), f"window {w} target {k}: {nrow} rows, oracle {len(want_top)}"
        got_top = top[off[k]:off[k + 1]]
        got_ls = ls[off[k]:off[k + 1]]
        assert np.all(got_top[:, N:] == 0), "padding columns must be zero"
        g, wv = got_top[:, :N].astype(np.float64), want_top.astype(np.float64)
        assert np.isfinite(g).all() and np.isfinite(got_ls).all()
        assert np.array_equal(g == 0, wv == 0), f"window {w} target {k}: zero pattern differs"
        assert (g[:, k] == 0).all()
        rel = np.abs(g - wv) / np.maximum(np.abs(wv), 1e-300)
        assert rel.max() <= REL_TOL, f"window {w} target {k}: topology max rel err {rel.max():.3e}"
        gl, wl = got_ls.astype(np.float64), want_ls.astype(np.float64)
        assert np.all(np.abs(gl - wl) <= REL_TOL * np.maximum(1.0, np.abs(wl))), f"window {w} target {k}: logscales differ"
        nz = wv != 0
        lp_g = np.log(g[nz]) + np.broadcast_to(gl[:, None], g.shape)[nz]
        lp_w = np.log(wv[nz]) + np.broadcast_to(wl[:, None], wv.shape)[nz]
        assert np.all(np.abs(lp_g - lp_w) <= REL_TOL * np.maximum(1.0, np.abs(lp_w)))
        nbad += np.count_nonzero(got_top[:, :N].view(np.uint32) != want_top.view(np.uint32))
        total += want_top.size
    return nbad / max(total, 1)


@pytest.mark.parametrize("N,L,W,seed,mode,theta,waves", [
    (70, 1500, 4, 41, "flat", 0.001, 0),
    (70, 1500, 4, 42, "skew", 0.001, 0),     # clamp 0.99 and rescales inside sections
    (130, 900, 3, 43, "flat", 0.025, 2),     # partial last word, two waves per target
    (300, 1200, 5, 44, "flat", 0.001, 0),
    (513, 700, 2, 45, "skew", 0.01, 4),
])
def test_repaint_matches_oracle(N, L, W, seed, mode, theta, waves):
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=seed, r_mode=mode, empty_window=(W >= 4))
    host_stones, res = gpu_paint_and_repaint(seq, r, theta, wb, range(W), waves)
    for w in range(W):
        top, ls, off, pitch = res[w]
        targets = None if N <= 130 else list(range(0, N, 7)) + [N - 1]
        frac = check_window(seq, r, theta, host_stones, w, top, ls, off, pitch, targets)
        _record(f"re-painting N={N} L={L} seed={seed} waves={waves} window {w}: share of floats not bit-identical {frac:.3e}")
        assert frac <= MAX_SHARE_NOT_IDENTICAL, f"{frac:.2e} of the floats are not bit-identical to the oracle"


@pytest.mark.parametrize("N,L,W,seed,mode,waves,geometry", [
    (1500, 400, 2, 46, "skew", 1, (1, 24, 2)),   # one wave, blocks of two rows
    (2100, 400, 3, 47, "flat", 2, (2, 20, 3)),   # 33 words over two waves
    (2100, 300, 2, 48, "skew", 1, (1, 36, 1)),   # every alpha row a checkpoint
    (4500, 200, 2, 49, "flat", 2, (2, 36, 1)),   # ... with two waves per target
    (1000, 600, 3, 50, "skew", 4, (4, 4, 8)),    # blocks of eight rows
    (4000, 200, 2, 51, "flat", 8, (8, 8, 8)),    # eight waves per target
])
def test_repaint_geometries(N, L, W, seed, mode, waves, geometry):
    """Wave counts, words per lane and block lengths that the benchmark's two windows do not reach, sampled targets
    (first, last, word and wave edges) against the pinned oracle bit for bit."""
    import torch
    ra = _ra()
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=seed, r_mode=mode)
    plan = ra.Plan(0, 1, waves)
    plan.load_host(seq, r, wb)
    plan.prepare(0.001, 0.001, 0, N)
    g = plan.repaint_geometry()
    plan.close()
    assert (g["waves_per_target"], g["states_per_lane"], g["checkpoint_rows"]) == geometry
    host_stones, res = gpu_paint_and_repaint(seq, r, 0.001, wb, range(W), waves)
    per_wave = 64 * geometry[1]
    targets = sorted(t for t in {0, 1, 63, 64, per_wave - 1, per_wave, N // 2, N - 2, N - 1} if t < N)
    for w in range(W):
        top, ls, off, pitch = res[w]
        frac = check_window(seq, r, 0.001, host_stones, w, top, ls, off, pitch, targets)
        _record(f"re-painting N={N} L={L} seed={seed} waves={waves} {geometry} window {w}: share of floats not bit-identical {frac:.3e}")
        assert frac <= MAX_SHARE_NOT_IDENTICAL, f"{frac:.2e} of the floats are not bit-identical to the oracle"


def test_repaint_known_answer_toy():
    """test/test_painting.cpp:98-133 through the GPU: with r = 0 the posterior is constant along the sequence
    and its log, in units of fast_log(theta/(1-theta)), is the asymmetric mismatch count d[k][n]."""
    seq, d = sy.toy_test_painting()
    N, L, theta = 5, 10, 0.025
    r = np.zeros(L)
    wb = np.array([0, L], dtype=np.int32)
    host_stones, res = gpu_paint_and_repaint(seq, r, theta, wb, [0])
    top, ls, off, pitch = res[0]
    o = ol.oracle()
    rescale = o.fast_log(theta / (1.0 - theta))
    for k in range(N):
        t, l = top[off[k]:off[k + 1], :N], ls[off[k]:off[k + 1]]
        norm = np.float32(o.fast_log(N - 1.0)) - np.float32(len(t)) * np.float32(o.fast_log(1 - theta))
        for j in range(len(t)):
            assert abs(l[0] - l[j]) < 1e-5
            for n in range(N):
                assert abs(t[j, n] - t[0, n]) < 1e-5
                if n != k:
                    v = (np.float32(o.fast_log(float(t[j, n]))) + l[j] + norm) / np.float32(rescale)
                    assert d[k, n] == round(float(v))
