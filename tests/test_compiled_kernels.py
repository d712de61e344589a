"""The painting kernels as COMPILED -- the gfx950 assembly the build keeps of every register-resident kernel file
(relate_amd/csrc/paint_regs_g*.s: compiler-generated code and the hand-written instruction streams inside it) --
interpreted on the CPU (tools/gfx950_emu.py) against the reference's records: every global, scalar, flat, scratch
and LDS access inside its buffer, no scalar register used before its load has been waited for, every stored float
the reference's.  This is what stands in for a GPU where there is none; the GPU tests say the same of the hardware."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle_lib as ol  # noqa: E402
import synth_np as sy  # noqa: E402
from conftest import load_golden  # noqa: E402

CSRC = os.path.join(ROOT, "relate_amd", "csrc")
pytestmark = pytest.mark.skipif(len(glob.glob(os.path.join(CSRC, "paint_regs_g*.s"))) < 4,
                                reason="no device assembly here: the build (make -C relate_amd/csrc) keeps it")


def same_records(got, want):
    g, w = ol.decode_records(got), ol.decode_records(want)
    for key in g:
        a, b = g[key], w[key]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), key


@pytest.mark.parametrize("name,targets", [("n130_tail", (0, 2)), ("n130_tail", (128, 130)), ("n70_skew_emptywin", (63, 66)),
                                          ("toy_r001", (0, 5))])
def test_one_wave_kernel_on_golden_chunks(name, targets):
    import emu_kernel as ek
    g = load_golden(name)
    t = range(*targets)
    out, stats = ek.paint_compiled(g["seq"], g["r"], g["wb"], g["theta_painter"], g["theta_data"], t, 1, 4)
    same_records(out, g["stones"][:, targets[0]:targets[1]])
    assert sum(stats.values()) > 1000


def test_two_theta_kernel_on_its_golden_chunk():
    """FastPainting(N, theta) beside Data::theta (the Rust caller): the two-row backward pass, which also executes
    the compiler's scratch spills."""
    import emu_kernel as ek
    g = load_golden("n33_mixed_theta")
    assert g["theta_painter"] != g["theta_data"]
    out, _ = ek.paint_compiled(g["seq"], g["r"], g["wb"], g["theta_painter"], g["theta_data"], range(30, 33), 1, 4)
    same_records(out, g["stones"][:, 30:33])


@pytest.mark.parametrize("N,G,MPL,tp,td", [(600, 2, 8, 0.001, 0.001), (600, 4, 4, 0.001, 0.001), (1100, 1, 20, 0.001, 0.001),
                                           (600, 2, 8, 0.001, 0.01), (600, 4, 4, 0.001, 0.01), (4000, 8, 8, 0.001, 0.001),
                                           (5008, 4, 20, 0.001, 0.001), (5008, 2, 40, 0.001, 0.001)])
def test_several_waves_and_several_rounds(N, G, MPL, tp, td):
    """Workgroups of several waves (LDS exchange through lanes 63 / 62, barrier per site), lanes of more words than a
    round of row words holds (two rounds per site), one and two thetas, the benchmark's 5008-haplotype geometry (4 x 20
    words: one round) and its two-wave alternative, against the oracle; targets at both ends of the panel and across
    a word."""
    import emu_kernel as ek
    seq, r, wb = sy.make_chunk(N=N, L=120, W=2, seed=N + G, r_mode="skew")
    for k0 in (0, 63, N - 2):
        want = ol.oracle().paint_stones(seq, r, tp, td, wb, k0, k0 + 2)
        out, _ = ek.paint_compiled(seq, r, wb, tp, td, range(k0, k0 + 2), G, MPL)
        same_records(out, want)


@pytest.mark.parametrize("N,L,G,MPL,k0,theta,mode", [(130, 400, 1, 4, 127, 0.025, "skew"), (600, 300, 2, 8, 62, 0.001, "flat"),
                                                     (1100, 200, 1, 20, 0, 0.001, "skew"), (2000, 300, 2, 16, 1998, 0.001, "flat"),
                                                     (4900, 200, 4, 20, 0, 0.001, "skew"),
                                                     (2040, 200, 1, 32, 2038, 0.001, "flat")])  # 32 words per lane: every alpha row a checkpoint
def test_repainting_kernels_from_checkpoints(N, L, G, MPL, k0, theta, mode):
    """repaint_fwd_kernel_r / repaint_bwd_kernel_r as compiled: the forward launch keeps every B-th alpha row, the
    backward launch recomputes the others into accumulation registers (the interpreter keeps them apart from the
    compiler's own and checks every access); topology rows and logscales against the re-painting oracle bit for bit.
    Geometries of the benchmark's windows among them (2 x 16 words: N = 2000; 4 x 20: N = 5008), several blocks per
    section, rescales inside sections (skewed map), and one geometry of 32 words per lane, where there is no room for
    slots and every alpha row is a checkpoint."""
    import emu_kernel as ek
    K = 2
    seq, r, wb = sy.make_chunk(N=N, L=L, W=2, seed=N, r_mode=mode)
    stones = ol.oracle().paint_stones(seq, r, theta, theta, wb, k0, k0 + K)
    blocks = 0
    for w in range(2):
        top, ls, off, _ = ek.repaint_compiled(seq, r, wb, theta, w, stones[w], range(k0, k0 + K), G, MPL)
        d = ol.decode_records(stones[w:w + 1])
        for t in range(K):
            want_top, want_ls = ol.oracle().repaint_section(seq, r, theta, int(d["a_snp"][0, t]), int(d["b_snp"][0, t]),
                                                            d["alpha"][0, t], d["beta"][0, t], float(d["a_ls"][0, t]),
                                                            float(d["b_ls"][0, t]), k0 + t)
            got_top, got_ls = top[off[t]:off[t + 1]], ls[off[t]:off[t + 1]]
            assert len(want_top) == off[t + 1] - off[t]
            assert np.array_equal(got_top[:, :N].view(np.uint32), want_top.view(np.uint32))
            assert np.array_equal(got_ls.view(np.uint32), want_ls.view(np.uint32))
            assert np.all(got_top[:, N:] == 0)
            blocks = max(blocks, -(-len(want_top) // ek.repaint_block_rows(MPL)))
    assert blocks > 1, "sections of several blocks"
    assert (ek.repaint_block_rows(MPL) == 1) == (MPL >= 32), "the one case of 32 words per lane keeps every row"


# ---- the plain kernels (paint_kernels.s) ------------------------------------------------------------------------
PLAIN = os.path.join(CSRC, "paint_kernels.s")
plain = pytest.mark.skipif(not os.path.exists(PLAIN), reason="no device assembly of the plain kernels here")


def _p(x):
    import struct
    return struct.pack("<Q", x)


def _i(x):
    import struct
    return struct.pack("<i", x)


def _f(x):
    import struct
    return struct.pack("<f", x)


def _d(x):
    import struct
    return struct.pack("<d", x)


@plain
@pytest.mark.parametrize("N,seed", [(5, 0), (70, 1), (130, 2)])
def test_minmatch_scan_kernels(N, seed):
    import gfx950_emu as emu
    import treebuilder_lib as tb
    d = tb.distance_like_matrix(N, seed)
    thr = tb.threshold(0.001)
    want = tb.oracle_scan(d, thr)
    cap = len(want[1]) + 3
    mem = emu.Memory()
    dm = mem.add("matrix", d.tobytes())
    mv = mem.add("min_values", bytes(4 * N), writable=True)
    cnt = mem.add("counts", bytes(4 * N), writable=True)
    off = mem.add("offsets", bytes(8 * (N + 1)), writable=True)
    ci, cj, cd = (mem.add(n, bytes(4 * cap), writable=True) for n in ("ci", "cj", "cd"))
    emu.launch(PLAIN, r"minmatch14row_min_kernel", mem, [_p(dm), _i(N), _f(thr), _p(mv)], (N, 1, 1), (64, 1, 1))
    emu.launch(PLAIN, r"minmatch23count_candidates_kernel", mem, [_p(dm), _p(mv), _i(N), _p(cnt)], (N, 1, 1), (64, 1, 1))
    emu.launch(PLAIN, r"minmatch18scan_counts_kernel", mem, [_p(cnt), _i(N), _p(off)], (1, 1, 1), (256, 1, 1))
    emu.launch(PLAIN, r"minmatch22fill_candidates_kernel", mem, [_p(dm), _p(mv), _i(N), _p(off), _p(cap), _p(ci), _p(cj), _p(cd)],
               (N, 1, 1), (64, 1, 1))
    n = int(np.frombuffer(bytes(mem.buffer("offsets")), dtype=np.int64)[N])
    assert n == len(want[1])
    assert np.array_equal(np.frombuffer(bytes(mem.buffer("min_values")), dtype=np.uint32), want[0].view(np.uint32))
    assert np.array_equal(np.frombuffer(bytes(mem.buffer("ci")), dtype=np.int32)[:n], want[1])
    assert np.array_equal(np.frombuffer(bytes(mem.buffer("cj")), dtype=np.int32)[:n], want[2])
    assert np.array_equal(np.frombuffer(bytes(mem.buffer("cd")), dtype=np.uint32)[:n], want[3].view(np.uint32))


@plain
@pytest.mark.parametrize("N,items", [(70, 4), (300, 4), (300, 0), (1100, 8)])
def test_distance_matrix_kernel(N, items):
    """distance_matrix_kernel<ITEMS> on a random topology: direct and mixed rows, both orders of the two logscales."""
    import gfx950_emu as emu
    rng = np.random.default_rng(N + items)
    rows, pitch = 2 * N + 3, (N + 63) // 64 * 64
    top = np.zeros((rows, pitch), dtype=np.float32)
    top[:, :N] = np.exp(rng.uniform(-30, 2, (rows, N))).astype(np.float32)
    ls = rng.uniform(-400, -350, rows).astype(np.float32)
    row = (2 * np.arange(N)).astype(np.int64)
    top[row, np.arange(N)] = 0.0  # a haplotype's own entry is zero in its topology rows
    interp = (rng.random(N) < 0.7).astype(np.uint8)
    wl = rng.random(N)
    wr = 1.0 - wl
    e_pn = np.exp((ls[row] - ls[row + 1]).astype(np.float64)).astype(np.float32)
    e_np = np.exp((ls[row + 1] - ls[row]).astype(np.float64)).astype(np.float32)
    want = ol.distance_matrix(top, ls, row, interp, wl, wr, N)
    mem = emu.Memory()
    a = [mem.add(n, x.tobytes()) for n, x in (("top", top), ("ls", ls), ("row", row), ("interp", interp), ("wl", wl), ("wr", wr),
                                              ("epn", e_pn), ("enp", e_np))]
    out = mem.add("matrix", bytes(4 * N * N), writable=True)
    nrun = N if N <= 300 else 96  # rows of the matrix (one workgroup each) that are run: all of a small one
    emu.launch(PLAIN, rf"distance_matrix_kernelILi{items}E", mem,
               [_p(a[0]), _i(pitch), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]), _i(N), _p(out)],
               (nrun, 1, 1), (256, 1, 1))
    got = np.frombuffer(bytes(mem.buffer("matrix")), dtype=np.float32).reshape(N, N)[:nrun]
    sel = slice(0, nrun)
    assert interp[sel].any() and not interp[sel].all() and (ls[row] <= ls[row + 1])[sel].any() and (ls[row] > ls[row + 1])[sel].any()
    assert np.array_equal(got.view(np.uint32), want[:nrun].view(np.uint32))


@plain
@pytest.mark.parametrize("N,L", [(130, 300), (70, 130)])
def test_prepare_kernels(N, L):
    """pack_rows, transpose, count_derived, fill_sites, site_terms, window_stones as compiled, chained as
    relate_plan_load_device / relate_plan_prepare chain them, against the per-target lists restated in
    tools/emu_paint_loop.py (bit rows, derived-site lists, window stones exact; the gap terms to the last bit or two:
    they pass through the device's exp)."""
    import struct
    import emu_paint_loop as lists
    import gfx950_emu as emu
    seq, r, wb = sy.make_chunk(N=N, L=L, W=3, seed=N, r_mode="skew")
    ch = lists.Chunk(seq, r, wb, 0.001)
    stride, cs = ch.stride, (L + 63) // 64
    mem = emu.Memory()
    # pack_rows reads the byte matrix in aligned 8-byte words: the last one may reach up to 7 bytes beyond L * N
    # (never beyond the page the last byte lies in, so no real allocation can end inside it)
    d_seq = mem.add("seq", seq.tobytes(), (seq.size + 7) // 8 * 8)
    d_rows = mem.add("rows", bytes(8 * L * stride), writable=True)
    d_cols = mem.add("cols", bytes(8 * (N * cs + 64)), writable=True)
    segs = L * (stride // 8)
    emu.launch(PLAIN, "pack_rows_kernel", mem, [_p(d_seq), _p(d_rows), _i(N), _i(L), _i(stride)], ((segs + 3) // 4, 1, 1), (256, 1, 1))
    rows = np.frombuffer(bytes(mem.buffer("rows")), dtype=np.uint64).reshape(L, stride)
    assert np.array_equal(rows, ch.rows)
    M = (N + 63) // 64
    emu.launch(PLAIN, "transpose_kernel", mem, [_p(d_rows), _p(d_cols), _i(N), _i(L), _i(stride), _i(cs)], (cs, (M + 3) // 4, 1), (256, 1, 1))
    cols = np.frombuffer(bytes(mem.buffer("cols")), dtype=np.uint64)[:N * cs].reshape(N, cs)
    bits = np.unpackbits(cols.view(np.uint8), axis=1, bitorder="little")[:, :L]
    assert np.array_equal(bits.T, (seq == ord("1")).astype(np.uint8))
    K = N
    d_nd = mem.add("nd", bytes(4 * K), writable=True)
    emu.launch(PLAIN, "count_derived_kernel", mem, [_p(d_cols), _i(cs), _i(L), _i(0), _i(K), _p(d_nd)], ((K + 3) // 4, 1, 1), (256, 1, 1))
    nd = np.frombuffer(bytes(mem.buffer("nd")), dtype=np.int32)
    assert list(nd) == [len(d) for d in ch.dk]
    off = np.concatenate([[0], np.cumsum(nd)[:-1]]).astype(np.int64)
    total = int(nd.sum())
    d_off = mem.add("off", off.tobytes())
    d_dk = mem.add("dk", bytes(4 * total), writable=True)
    emu.launch(PLAIN, "fill_sites_kernel", mem, [_p(d_cols), _i(cs), _i(L), _i(0), _i(K), _p(d_off), _p(d_dk)], ((K + 3) // 4, 1, 1), (256, 1, 1))
    dk = np.frombuffer(bytes(mem.buffer("dk")), dtype=np.int32)
    assert np.array_equal(dk, np.concatenate(ch.dk))
    d_r = mem.add("r", np.asarray(r, dtype=np.float64).tobytes())
    d_rec = mem.add("rec", bytes(24 * total), writable=True)
    c = ch.c
    bx = min(64, max(1, (int(nd.max()) + 255) // 256))
    emu.launch(PLAIN, "site_terms_kernel", mem, [_p(d_dk), _p(d_off), _p(d_nd), _p(d_r), _d(c["log_ntheta"]), _d(c["log_small"]),
                                                 _d(c["Nm1"]), _p(0), _p(0), _p(d_rec)], (bx, K, 1), (256, 1, 1))
    rec = bytes(mem.buffer("rec"))
    worst = 0.0
    for k in (0, 1, N // 2, N - 1):
        for j, (cf, nx, f2, b2) in enumerate(ch.rec[k]):
            gcf, gnx, gf2, gb2 = struct.unpack_from("<ddii", rec, 24 * (int(off[k]) + j))
            assert (gf2, gb2) == (f2, b2)
            worst = max(worst, abs(gcf - cf) / cf, abs(gnx - nx) / abs(nx))
    # 1 - exp(-x) amplifies the last bit of the device's exp where the gap is short (x small): ulps, not more
    assert worst < 1e-12
    W = len(wb) - 1
    d_wb = mem.add("wb", np.asarray(wb, dtype=np.int32).tobytes())
    d_bbj = mem.add("bbj", bytes(4 * K * W), writable=True)
    d_bej = mem.add("bej", bytes(4 * K * W), writable=True)
    emu.launch(PLAIN, "window_stones_kernel", mem, [_p(d_dk), _p(d_off), _p(d_nd), _i(K), _i(W), _p(d_wb), _p(d_bbj), _p(d_bej)],
               ((K * W + 255) // 256, 1, 1), (256, 1, 1))
    assert np.array_equal(np.frombuffer(bytes(mem.buffer("bbj")), dtype=np.int32).reshape(K, W), ch.bbj)
    assert np.array_equal(np.frombuffer(bytes(mem.buffer("bej")), dtype=np.int32).reshape(K, W), ch.bej)


@plain
def test_fast_log_kernel_matches_reference_bits():
    import gfx950_emu as emu
    from conftest import load_fast_log_golden
    x, want = load_fast_log_golden()
    x, want = x[::8][:2048], want[::8][:2048]
    mem = emu.Memory()
    a = mem.add("x", x.tobytes())
    b = mem.add("y", bytes(4 * len(x)), writable=True)
    emu.launch(PLAIN, "fast_log_kernel", mem, [_p(a), _p(b), _i(len(x))], ((len(x) + 255) // 256, 1, 1), (256, 1, 1))
    assert np.array_equal(np.frombuffer(bytes(mem.buffer("y")), dtype=np.uint32), want.view(np.uint32))
