"""CPU tests of the host side: chunk file formats, the window rule, the synthetic generator, the
command-line front end, and the sharding arithmetic."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
import relate_amd as ra
import synth_np as sy
from relate_amd import dist as rdist


def test_chunk_files_round_trip(tmp_path):
    """Formats of Data::MakeChunks (src/data.cpp:272-286,:496-526), read back with numpy."""
    seq, r, wb = sy.make_chunk(N=37, L=200, W=4, seed=1)
    ra.write_chunk(tmp_path, 3, seq, r, wb)
    N, L, wb2 = rdist.read_parameters(tmp_path, 3)
    assert (N, L) == (37, 200) and np.array_equal(wb2, wb)
    seq2, r2 = rdist.read_chunk(tmp_path, 3)
    assert np.array_equal(seq2, seq) and np.array_equal(r2, r)
    raw = np.fromfile(tmp_path / "parameters_c3.bin", dtype=np.int32)
    assert raw[2] == len(wb) and raw.size == 3 + len(wb)
    assert os.path.getsize(tmp_path / "chunk_3.hap") == 16 + 37 * 200
    bp = np.fromfile(tmp_path / "chunk_3.bp", dtype=np.int32)
    assert bp[0] == 200 and bp.size == 201
    rpos = np.fromfile(tmp_path / "chunk_3.rpos", dtype=np.uint8)
    assert rpos.size == 4 + 8 * 201  # uint32 L+1, then L+1 doubles
    head = np.frombuffer(rpos[:4].tobytes(), dtype=np.uint32)[0]
    pos = np.frombuffer(rpos[4:].tobytes(), dtype=np.float64)
    assert head == 201
    # r[l] = max(rpos[l+1]-rpos[l], 1e-10) * 2500  (src/data.cpp:484-488)
    assert np.allclose(np.maximum(np.diff(pos), 1e-10) * 2500, r, rtol=1e-9)
    state = np.fromfile(tmp_path / "chunk_3.state", dtype=np.int32)
    assert state[0] == 200 and state.size == 201
    # Data mirror reads the same files
    d = ra.Data(tmp_path / "chunk_3.hap", None, None, tmp_path / "chunk_3.r", None, None)
    assert (d.N, d.L) == (37, 200)
    d.SetPainting(0.01, 2.0)


def test_window_rule_matches_restatement():
    """src/data.cpp:145,:219-255: a window closes at SNP s when the running sum of derived*(N+1) reaches
    memory*1e9/4 - (2N^2+3N) floats and the window holds more than 10 SNPs."""
    seq, _ = ra.synth_chunk(120, 3000, 5)
    N, L = 120, 3000
    mem = 0.0006
    budget = mem * 1e9 / 4.0 - (2 * N * N + 3 * N)
    wb = [0]
    acc, inw = 0.0, 0
    for s in range(L):
        acc += int((seq[s] == ord("1")).sum()) * (N + 1)
        if acc >= budget and inw > 10:
            inw, acc = 0, 0.0
            wb.append(s)
        inw += 1
    wb.append(L)
    got = ra.window_boundaries(seq, mem)
    assert np.array_equal(got, np.array(wb, dtype=np.int32))
    assert len(got) > 3
    with pytest.raises(ra.RelateError):
        ra.window_boundaries(seq, 1e-7)  # "Need larger memory allowance"


def test_synth_chunk_is_deterministic_and_plausible():
    a, ra_ = ra.synth_chunk(300, 4000, 11)
    b, rb = ra.synth_chunk(300, 4000, 11)
    c, _ = ra.synth_chunk(300, 4000, 12)
    assert np.array_equal(a, b) and np.array_equal(ra_, rb) and not np.array_equal(a, c)
    assert set(np.unique(a)) <= {ord("0"), ord("1")}
    d = a == ord("1")
    cnt = d.sum(1)
    assert cnt.min() >= 1 and cnt.max() <= 299          # every SNP segregates
    expected = 1.0 / sum(1.0 / i for i in range(1, 300))  # mean derived frequency of the 1/i spectrum
    assert abs(d.mean() - expected) < 0.35 * expected
    # linkage disequilibrium: neighbours correlate more than distant sites
    x = d.astype(np.float64)
    x -= x.mean(1, keepdims=True)
    nrm = np.sqrt((x * x).sum(1))
    def r2(lag):
        v = (x[:-lag] * x[lag:]).sum(1) / (nrm[:-lag] * nrm[lag:])
        return float(np.mean(v * v))
    assert r2(1) > 3 * r2(1000)
    _, rs = ra.synth_chunk(50, 2000, 3, skewed_map=True)
    assert rs.min() >= 2.5e-7 and rs.max() <= 10.0 and rs.max() / rs.min() > 1e4
    assert ra_.min() > 0.012 and ra_.max() < 0.038


def test_cli_usage_and_errors(tmp_path):
    exe = os.path.join(os.path.dirname(ra.library_path()), "relate-gpu")
    assert os.path.exists(exe), "build the library first (make -C relate_amd/csrc)"
    p = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert p.returncode == 0 and "--mode Paint" in p.stderr
    p = subprocess.run([exe, "--mode", "BuildTopology", "-o", str(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 1
    p = subprocess.run([exe, "--mode", "Paint", "-o", str(tmp_path / "missing")], capture_output=True, text=True)
    assert p.returncode == 1 and "relate-gpu:" in p.stderr


@pytest.mark.parametrize("N,world", [(10, 1), (10, 3), (2000, 8), (5008, 8), (7, 8)])
def test_shard_ranges_cover_all_targets(N, world):
    ranges = [rdist.shard_range(N, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == N
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0
    sizes = [b - a for a, b in ranges]
    assert max(sizes) - min(sizes) <= 1
