"""The .haps / .sample / map loader (host side, no GPU): bit rows, positions, recombination distances and the
chunk / window cuts against a restatement of Data::MakeChunks written here in Python from the reference's rules
(src/data.cpp:134-528, src/data.hpp:115-149).  The reference's own reader needs gzstream.h and cannot be built
in this image, and the reference ships no data set: parity of this row is unpinned; what is checked is that two
independent restatements (C++ in the library, Python here) agree, and that the files the loader writes are the
ones the chunk readers of the painting stage accept."""
import gzip
import os
import struct

import numpy as np
import pytest

import relate_amd as ra
from relate_amd.haps import Haps, row_stride_words

ALLELES = "ACGT"


def write_data_set(d, seq, bp, rng, map_bp, map_cm, haploid_every=0, gz=False, trailing_newline=True):
    """seq: (L, N) uint8 of '0'/'1'.  Individuals are diploid (ID_1 == ID_2) except every haploid_every-th."""
    L, N = seq.shape
    opener = (lambda p: gzip.open(p, "wt")) if gz else (lambda p: open(p, "w"))
    ext = ".gz" if gz else ""
    haps, sample, gmap = (os.path.join(d, "x.haps" + ext), os.path.join(d, "x.sample" + ext), os.path.join(d, "x.map" + ext))
    anc = rng.integers(0, 4, L)
    alt = (anc + rng.integers(1, 4, L)) % 4
    with opener(haps) as f:
        for s in range(L):
            line = f"1 rs{s} {bp[s]} {ALLELES[anc[s]]} {ALLELES[alt[s]]} " + " ".join(chr(c) for c in seq[s])
            f.write(line + ("\n" if (s + 1 < L or trailing_newline) else ""))
    with opener(sample) as f:
        f.write("ID_1 ID_2 missing\n0 0 0\n")
        n, i = 0, 0
        while n < N:
            i += 1
            if (haploid_every and i % haploid_every == 0) or n + 1 == N:
                f.write(f"ind{i} NA 0\n")
                n += 1
            else:
                f.write(f"ind{i} ind{i} 0\n")
                n += 2
    with opener(gmap) as f:
        f.write("pos COMBINED_rate Genetic_Map\n")
        for b, c in zip(map_bp, map_cm):
            f.write(f"{int(b)} 1.0 {float(c)!r}\n")
    states = np.array([0 if {ALLELES[a], ALLELES[b]} in ({"C", "T"}, {"A", "G"}) else 1 for a, b in zip(anc, alt)],
                      dtype=np.int32)
    return haps, sample, gmap, states


def expected_rows(seq):
    L, N = seq.shape
    stride = row_stride_words(N)
    bits = np.ones((L, stride * 64), dtype=np.uint8)  # bits from N on are set
    bits[:, :N] = seq == ord("1")
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, stride)


def expected_rpos(map_bp, map_cm, bp_plus):
    """src/data.cpp:452-479, statement by statement."""
    mbp = [int(float(b)) for b in map_bp]
    rpos = [0.0] * len(bp_plus)
    i, mp = 0, 0
    if mbp[0] > bp_plus[0]:
        rpos[0] = map_cm[0] * 1e-2
        i = 1
    while i < len(bp_plus):
        while mbp[mp + 1] <= bp_plus[i] and mp < len(mbp) - 2:
            mp += 1
        if mbp[mp + 1] - mbp[mp] == 0 or mbp[mp] > bp_plus[i]:
            rpos[i] = map_cm[mp] * 1e-2
        else:
            rpos[i] = ((bp_plus[i] - mbp[mp]) / float(mbp[mp + 1] - mbp[mp]) * (map_cm[mp + 1] - map_cm[mp]) + map_cm[mp]) * 1e-2
        i += 1
    return np.array(rpos)


def expected_chunks(seq, memory_gb):
    """src/data.cpp:145-267,:288-313: list of (first_snp, L_chunk, boundaries relative to first_snp)."""
    L, N = seq.shape
    derived = (seq == ord("1")).sum(axis=1)
    budget = float(np.float32(memory_gb)) * 1e9 / 4.0 - (2 * N * N + 3 * N)
    assert budget > 0
    max_chunk = 2500000 if memory_gb >= 100 else min(L + 1, int(budget / N))
    out, snp, wb, nw, starts = [], 0, [0] * 501, 0, [0]
    while snp < L:
        over = []
        if snp > 0:
            assert snp - starts[-1] >= 20000
            begin = snp - 20000
            starts.append(begin)
            over = [begin] + [b for b in wb[:nw] if b > begin]
            assert len(over) < 499
        snp_begin, mem, size, inw = snp, 0.0, 0, 0
        wb[0], nw = snp, 1
        while nw + len(over) < 500 and size < max_chunk and snp < L:
            mem += int(derived[snp]) * (N + 1)
            if mem >= budget and inw > 10:
                inw, mem = 0, 0.0
                wb[nw] = snp
                nw += 1
            snp += 1
            inw += 1
            size += 1
        wb[nw] = snp
        if snp_begin == 0:
            out.append((0, size, list(wb[:nw + 1])))
        else:
            out.append((over[0], size + 20000, [b - over[0] for b in over] + [b - over[0] for b in wb[:nw + 1]]))
    return out


def memory_for_budget(budget, N):
    return (budget + 2 * N * N + 3 * N) * 4.0 / 1e9


def make_case(tmp_path, N, L, seed, budget, **kw):
    rng = np.random.default_rng(seed)
    seq, _ = ra.synth_chunk(N, L, seed)
    bp = np.cumsum(rng.integers(1, 400, L)).astype(np.int64) + 5000
    nmap = 40
    map_bp = np.sort(rng.choice(np.arange(int(bp[0]) + 3000, int(bp[-1]) - 3000), nmap, replace=False))
    map_bp[7] = map_bp[6]  # an interval of length zero (src/data.cpp:471)
    map_cm = np.cumsum(rng.random(nmap) * 0.02)
    paths = write_data_set(str(tmp_path), seq, bp, rng, map_bp, map_cm, **kw)
    return seq, bp, map_bp, map_cm, paths, memory_for_budget(budget, N)


def check_against_restatement(h, seq, bp, map_bp, map_cm, mem):
    L, N = seq.shape
    assert (h.N, h.L) == (N, L)
    assert np.array_equal(h.rows(), expected_rows(seq))
    gbp, gdist, grpos, gr = h.positions()
    bp_plus = np.append(bp, bp[-1] + 1)
    assert np.array_equal(gbp, bp_plus)
    assert np.array_equal(gdist, np.append(np.diff(bp), 1))
    want_rpos = expected_rpos(map_bp, list(map_cm), [int(b) for b in bp_plus])
    assert np.array_equal(grpos, want_rpos)  # same operations in the same order: the same doubles
    assert np.array_equal(gr, np.maximum(np.diff(want_rpos), 1e-10) * 2500)
    want = expected_chunks(seq, mem)
    assert h.num_chunks == len(want)
    for c, (first, lc, wb) in enumerate(want):
        gf, gl, gwb = h.chunk(c)
        assert (gf, gl) == (first, lc) and list(gwb) == wb, f"chunk {c}"
    return want


def test_one_chunk_many_windows_against_the_restatement(tmp_path):
    seq, bp, map_bp, map_cm, (haps, sample, gmap, _), mem = make_case(tmp_path, N=70, L=3000, seed=3, budget=70 * 3001.0,
                                                                       haploid_every=3)
    with Haps(haps, sample, gmap, mem) as h:
        want = check_against_restatement(h, seq, bp, map_bp, map_cm, mem)
        assert len(want) == 1 and len(want[0][2]) > 5
        # the existing one-chunk window rule (what relate_write_chunk's callers use) says the same
        assert list(ra.window_boundaries(seq, memory_gb=mem)) == want[0][2]


def test_map_that_starts_before_and_ends_after_the_snps(tmp_path):
    rng = np.random.default_rng(5)
    seq, _ = ra.synth_chunk(16, 400, 5)
    bp = np.cumsum(rng.integers(1, 90, 400)) + 100000
    map_bp = np.array([10, 50000, 100500, 104000, 110000, 400000])
    map_cm = np.array([0.0, 0.3, 0.31, 0.5, 0.5, 2.5])
    haps, sample, gmap, _ = write_data_set(str(tmp_path), seq, bp, rng, map_bp, map_cm)
    with Haps(haps, sample, gmap, 5.0) as h:
        check_against_restatement(h, seq, bp, map_bp, map_cm, 5.0)
        _, _, _, r = h.positions()
        assert (r >= 2500 * 1e-10).all() and (r == 2500 * 1e-10).any()  # the flat stretch of the map hits the floor


def test_chunks_overlap_by_20000_snps_and_inherit_windows(tmp_path):
    N, L = 8, 47000
    seq, bp, map_bp, map_cm, (haps, sample, gmap, states), mem = make_case(tmp_path, N, L, seed=9, budget=N * 26000.0,
                                                                            gz=True)
    with Haps(haps, sample, gmap, mem) as h:
        want = check_against_restatement(h, seq, bp, map_bp, map_cm, mem)
        assert len(want) == 2 and want[1][0] == want[0][1] - 20000 and want[1][0] + want[1][1] == L
        assert len(want[1][2]) >= 3  # the windows inside the overlap come first
        out = tmp_path / "chunks"
        out.mkdir()
        h.write_chunks(str(out), use_transitions=False)
        _, _, rpos, r = h.positions()
    # parameters.bin: N, L, chunks, memory, first and last SNP of every chunk (src/data.cpp:380-391)
    raw = (out / "parameters.bin").read_bytes()
    n, l, nc = struct.unpack_from("<iii", raw, 0)
    assert (n, l, nc) == (N, L, 2) and len(raw) == 12 + 8 + 8 * nc
    assert list(struct.unpack_from("<4i", raw, 20)) == [0, want[1][0], want[0][1], L]
    for c, (first, lc, wb) in enumerate(want):
        raw = (out / f"parameters_c{c}.bin").read_bytes()
        assert list(struct.unpack_from(f"<{3 + len(wb)}i", raw, 0)) == [N, lc, len(wb)] + wb
        hap = np.fromfile(out / f"chunk_{c}.hap", dtype=np.uint8)
        assert list(struct.unpack_from("<QQ", hap[:16].tobytes(), 0)) == [lc, N]
        assert np.array_equal(hap[16:].reshape(lc, N), seq[first:first + lc])
        for ext, dt, vec in ((".bp", np.int32, bp[first:first + lc]), (".r", np.float64, r[first:first + lc]),
                             (".rpos", np.float64, rpos[first:first + lc + 1]), (".dist", np.int32, np.append(np.diff(bp), 1)[first:first + lc])):
            v = np.fromfile(out / f"chunk_{c}{ext}", dtype=np.uint8)
            assert struct.unpack_from("<I", v[:4].tobytes(), 0)[0] == len(vec)
            assert np.array_equal(v[4:].view(dt), vec.astype(dt))
        st = np.fromfile(out / f"chunk_{c}.state", dtype=np.int32)
        assert st[0] == lc and np.array_equal(st[1:], states[first:first + lc])
    props = (out / "props.bin").read_bytes()
    assert len(props) == L * (12 + 3 * 1024)
    s = 12345
    rec = props[s * 3084:(s + 1) * 3084]
    assert struct.unpack_from("<iii", rec, 0) == (s, int(bp[s]), int(bp[s + 1] - bp[s]))
    assert rec[12:12 + 1024].rstrip(b"\0") == f"rs{s}".encode()


def test_written_chunk_is_what_the_painting_stage_reads(tmp_path):
    """chunk files of the loader through the library's own chunk reader (Data::Data, src/data.cpp:88-99)."""
    seq, bp, map_bp, map_cm, (haps, sample, gmap, _), mem = make_case(tmp_path, N=33, L=500, seed=11, budget=33 * 501.0)
    out = tmp_path / "chunks"
    out.mkdir()
    with Haps(haps, sample, gmap, mem) as h:
        h.write_chunks(str(out))
        _, _, _, r = h.positions()
    d = ra.Data(str(out / "chunk_0.hap"), str(out / "chunk_0.bp"), str(out / "chunk_0.dist"), str(out / "chunk_0.r"),
                str(out / "chunk_0.rpos"), str(out / "chunk_0.state"))
    assert (d.N, d.L) == (33, 500)


def test_last_line_without_newline_is_not_a_snp(tmp_path):
    rng = np.random.default_rng(2)
    seq, _ = ra.synth_chunk(6, 30, 2)
    bp = np.arange(30) * 10 + 7
    haps, sample, gmap, _ = write_data_set(str(tmp_path), seq, bp, rng, [1, 1000], [0.0, 1.0], trailing_newline=False)
    with Haps(haps, sample, gmap, 5.0) as h:
        assert (h.N, h.L) == (6, 29)  # L counts newlines (src/data.hpp:136-141)


@pytest.mark.parametrize("what", ["unsorted", "short_line", "memory", "missing", "one_map_line"])
def test_malformed_input_is_refused_with_a_message(tmp_path, what):
    rng = np.random.default_rng(4)
    seq, _ = ra.synth_chunk(10, 40, 4)
    bp = np.arange(40) * 10 + 7
    mem = 5.0
    map_bp, map_cm = [1, 1000], [0.0, 1.0]
    if what == "unsorted":
        bp[20] = bp[19]
    if what == "one_map_line":
        map_bp, map_cm = [1], [0.0]
    haps, sample, gmap, _ = write_data_set(str(tmp_path), seq, bp, rng, map_bp, map_cm)
    if what == "short_line":
        lines = open(haps).read().split("\n")
        lines[5] = lines[5][:-2]
        open(haps, "w").write("\n".join(lines))
    if what == "memory":
        mem = 1e-7
    if what == "missing":
        gmap = gmap + ".absent"
    with pytest.raises(ra.RelateError) as e:
        Haps(haps, sample, gmap, mem)
    assert str(e.value).split(": ", 1)[1].strip()
    assert e.value.code == {"unsorted": -1, "short_line": -1, "memory": -5, "missing": -2, "one_map_line": -1}[what]


def test_command_line_makechunks_writes_the_same_files(tmp_path):
    """relate-gpu --mode MakeChunks (options of pipeline/MakeChunks.cpp:15-72) against the library call."""
    import subprocess
    exe = os.path.join(os.path.dirname(ra.library_path()), "relate-gpu")
    if not os.path.exists(exe):
        pytest.skip("relate-gpu has not been built")
    seq, bp, map_bp, map_cm, (haps, sample, gmap, states), mem = make_case(tmp_path, N=20, L=300, seed=13, budget=20 * 301.0)
    a, b = tmp_path / "by_cli", tmp_path / "by_call"
    cmd = [exe, "--mode", "MakeChunks", "--haps", haps, "--sample", sample, "--map", gmap, "-o", str(a), "--memory", repr(mem),
           "--transversion"]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stderr
    b.mkdir()
    with Haps(haps, sample, gmap, mem) as h:
        h.write_chunks(str(b), use_transitions=False)
    names = sorted(os.listdir(a))
    assert names == sorted(os.listdir(b)) and "chunk_0.hap" in names and "parameters.bin" in names and "props.bin" in names
    for n in names:
        assert (a / n).read_bytes() == (b / n).read_bytes(), n
    st = np.fromfile(a / "chunk_0.state", dtype=np.int32)
    assert np.array_equal(st[1:], states)
    # an existing directory is refused, as the reference refuses it
    again = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert again.returncode == 1 and "already exists" in again.stderr


def test_any_separator_between_alleles(tmp_path):
    """haps::ReadSNP counts '0' and '1' and skips every other character (src/data.cpp:565-577): tabs, runs of
    blanks, carriage returns and alleles with nothing between them give the same bit rows as single blanks (the
    parser's four-at-a-time path takes only the usual " 0 1" text and must hand over cleanly in the middle of a line)."""
    rng = np.random.default_rng(21)
    N, L = 131, 60
    seq, _ = ra.synth_chunk(N, L, 21)
    bp = np.arange(L) * 10 + 7
    haps, sample, gmap, _ = write_data_set(str(tmp_path), seq, bp, rng, [1, 1000], [0.0, 1.0], haploid_every=2)
    seps = [" ", "  ", "\t", "", " \t ", "x"]
    with open(haps, "w") as f:
        for s_ in range(L):
            parts = [f"1 rs{s_} {bp[s_]} A G"]
            for n in range(N):
                # long stretches of the usual text, with something else now and then
                sep = " " if (n == 0 or rng.random() < 0.9) else seps[rng.integers(0, len(seps))]  # the first allele follows the fifth field
                parts.append(sep + chr(seq[s_, n]))
            f.write("".join(parts) + ("\r\n" if s_ % 3 == 0 else "\n"))
    with Haps(haps, sample, gmap, 5.0) as h:
        assert (h.N, h.L) == (N, L)
        assert np.array_equal(h.rows(), expected_rows(seq))
