"""CPU tests that pin the oracle (oracle/paint_oracle.c): against the golden
vectors produced by the compiled reference, against the reference's own
known-answer tests, and live against oracle/_ref where that library exists."""
import math

import numpy as np
import pytest

import oracle_lib as ol
import synth_np as sy
from conftest import load_golden


def test_oracle_matches_golden_stones_bytewise(golden):
    g = golden
    out = ol.oracle().paint_stones(g["seq"], g["r"], g["theta_painter"], g["theta_data"], g["wb"])
    assert out.shape == g["stones"].shape
    assert np.array_equal(out, g["stones"])


def test_oracle_matches_golden_repaint(golden):
    g = golden
    if "rp0_wk" not in g:
        pytest.skip("mixed-theta case has no RePaintSection fixture")
    st = ol.decode_records(g["stones"])
    for i in range(4):
        w, k = (int(v) for v in g[f"rp{i}_wk"])
        top, ls = ol.oracle().repaint_section(g["seq"], g["r"], g["theta_data"], int(st["a_snp"][w, k]),
                                              int(st["b_snp"][w, k]), st["alpha"][w, k], st["beta"][w, k],
                                              float(st["a_ls"][w, k]), float(st["b_ls"][w, k]), k)
        assert np.array_equal(top.view(np.uint32), g[f"rp{i}_top"].view(np.uint32))
        assert np.array_equal(ls.view(np.uint32), g[f"rp{i}_ls"].view(np.uint32))


def test_record_layout_discrete_fields(golden):
    """Record = i32 start, i32 end, {u64 1, u64 N, i32 snp, f32 ls, f32[N]} x 2
    (src/fast_painting.cpp:603-615, src/collapsed_matrix.hpp:240-252)."""
    g = golden
    d = ol.decode_records(g["stones"])
    W = len(g["wb"]) - 1
    N = g["N"]
    assert g["stones"].shape == (W, N, 56 + 8 * N)
    for w in range(W):
        assert (d["start"][w] == g["wb"][w]).all() and (d["end"][w] == g["wb"][w + 1] - 1).all()
    assert (d["a_rows"] == 1).all() and (d["b_rows"] == 1).all()
    assert (d["a_cols"] == N).all() and (d["b_cols"] == N).all()
    assert (d["a_snp"][0] == 0).all() and (d["b_snp"][W - 1] == g["L"] - 1).all()
    assert (d["a_snp"] < d["b_snp"]).all()
    # alpha[k] is 0 in every stone; beta[k] is 0 except in stones taken at the last SNP, where it is 1
    for k in range(N):
        assert (d["alpha"][:, k, k] == 0).all()
        at_end = d["b_snp"][:, k] == g["L"] - 1
        assert (d["beta"][at_end, k, k] == 1).all() and (d["beta"][~at_end, k, k] == 0).all()


def test_fast_log_error_bound():
    """test/test_log.cpp:6-15: |fast_log(x) - log(x)| < 0.007 over decades 1e-4..1e4."""
    fl = ol.oracle().fast_log
    for e in range(-4, 5):
        x = 10.0 ** e
        assert abs(fl(x) - math.log(x)) < 0.007
    xs = np.exp(np.random.default_rng(1).uniform(math.log(1e-30), math.log(1e30), 2000))
    assert max(abs(fl(float(np.float32(x))) - math.log(float(np.float32(x)))) for x in xs) < 0.007


@pytest.mark.parametrize("name", ["toy_r0"])
def test_known_answer_mismatch_counts(name):
    """Restates test/test_painting.cpp:98-133 on the oracle: with r = 0 the
    posterior is constant along the sequence and its log, in units of
    fast_log(theta/(1-theta)), is the asymmetric mismatch count d[k][n]."""
    g = load_golden(name)
    o = ol.oracle()
    seq, d = sy.toy_test_painting()
    assert np.array_equal(seq, g["seq"]) and np.array_equal(d, g["expected_d"])
    N, L, theta = 5, 10, 0.025
    r = np.zeros(L)
    pt = np.float32(theta / (N - 1.0) - (1 - theta) / (N - 1.0))
    pn = np.float32((1 - theta) / (N - 1.0))
    rescale = o.fast_log(theta / (1.0 - theta))
    for k in range(N):
        ab = np.array([np.float32(float(seq[0, k] > seq[0, n])) * pt + pn for n in range(N)], dtype=np.float32)
        top, ls = o.repaint_section(seq, r, theta, 0, L - 1, ab, np.ones(N, np.float32), 0.0, 0.0, k)
        assert np.array_equal(top.view(np.uint32), g[f"toy_top{k}"].view(np.uint32))
        assert np.array_equal(ls.view(np.uint32), g[f"toy_ls{k}"].view(np.uint32))
        norm = np.float32(o.fast_log(N - 1.0)) - np.float32(len(top)) * np.float32(o.fast_log(1 - theta))
        for l in range(len(top)):
            assert abs(ls[0] - ls[l]) < 1e-5
            for n in range(N):
                assert abs(top[l, n] - top[0, n]) < 1e-5
                if n != k:
                    v = (np.float32(o.fast_log(float(top[l, n]))) + ls[l] + norm) / np.float32(rescale)
                    assert d[k, n] == round(float(v))


@pytest.mark.skipif(not ol.ref_available(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_oracle_matches_compiled_reference_live(seed):
    seq, r, wb = sy.make_chunk(N=20 + 7 * seed, L=600, W=4, seed=seed, r_mode="skew" if seed % 2 else "flat")
    for tp, td in ((0.001, 0.001), (0.001, 0.01)):
        a = ol.oracle().paint_stones(seq, r, tp, td, wb)
        b = ol.ref().paint_stones(seq, r, tp, td, wb)
        assert np.array_equal(a, b)


def test_distance_oracle_against_numpy_restatement():
    """oracle_distance_matrix (src/anc_builder.cpp:120-196 restated in C; parity unpinned, see its header) against
    the same expressions written with numpy and the oracle's fast_log."""
    rng = np.random.default_rng(3)
    N, rows, pitch = 9, 30, 16
    top = np.zeros((rows, pitch), dtype=np.float32)
    top[:, :N] = np.exp(rng.uniform(-60, 0, (rows, N))).astype(np.float32)
    ls = rng.uniform(-300, 5, rows).astype(np.float32)
    row = rng.integers(0, rows - 1, N).astype(np.int64)
    interp = (rng.random(N) < 0.5).astype(np.uint8)
    wl = rng.random(N)
    wr = 1.0 - wl
    for n in range(N):
        top[row[n], n] = 0.0  # a target's own entry
    got = ol.distance_matrix(top, ls, row, interp, wl, wr, N)
    fl = ol.oracle().fast_log
    f32 = np.float32
    for n in range(N):
        vals = np.zeros(N, dtype=np.float32)
        for m in range(N):
            tp, tn = top[row[n], m], top[row[n] + 1, m]
            lp, ln = ls[row[n]], ls[row[n] + 1]
            if not interp[n]:
                vals[m] = (f32(fl(float(tp))) + lp) * f32(-1.0)
            elif lp <= ln:
                x = wl[n] * float(tp) * float(f32(np.exp(np.float64(f32(lp - ln))))) + wr[n] * float(tn)
                vals[m] = (f32(fl(float(f32(x)))) + ln) * f32(-1.0)
            else:
                x = wl[n] * float(tp) + wr[n] * float(tn) * float(f32(np.exp(np.float64(f32(ln - lp)))))
                vals[m] = (f32(fl(float(f32(x)))) + lp) * f32(-1.0)
        mn = vals.min()
        want = vals - mn
        want[n] = 0.0
        assert np.array_equal(got[n].view(np.uint32), want.view(np.uint32)), n


def test_fast_log_matches_reference_bits():
    """oracle_fast_log against the bits the compiled reference produced (src/fast_log.hpp:6-21) for 80 000
    arguments: decades (test/test_log.cpp:6-15), powers of two and ten, boundary neighbours, denormals, random."""
    from conftest import load_fast_log_golden
    x, y = load_fast_log_golden()
    fl = ol.oracle().fast_log
    got = np.array([fl(float(v)) for v in x], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), y.view(np.uint32))
    if ol.ref_available():
        rf = ol.ref().fast_log
        again = np.array([rf(float(v)) for v in x[:5000]], dtype=np.float32)
        assert np.array_equal(again.view(np.uint32), y[:5000].view(np.uint32))
    ok = np.isfinite(x) & (x >= 1e-4) & (x <= 1e4)
    assert np.max(np.abs(got[ok].astype(np.float64) - np.log(x[ok].astype(np.float64)))) < 0.007  # test/test_log.cpp:7
