"""The window files appear complete or not at all (run with -m gpu).

Consumers of relate_<w>.bin trust its size, N records (src/anc_builder.cpp:65-69), and later stages delete their
inputs (pipeline/FindEquivalentBranches.cpp:43-60), so a run that fails half way must not leave a file of the full
size with a hole in it.  The failure is injected from outside: the painting runs in a child process whose file-size
limit (RLIMIT_FSIZE) is smaller than one window file, so a pwrite in the middle of the copy-back fails."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import oracle_lib as ol
import synth_np as sy

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_failed_write_leaves_no_window_file(tmp_path):
    import relate_amd as ra
    seq, r, wb = sy.make_chunk(N=90, L=2500, W=5, seed=41, r_mode="flat")
    ra.write_chunk(tmp_path, 0, seq, r, wb)
    warm = tmp_path / "warm"
    warm.mkdir()
    ra.write_chunk(warm, 0, seq, r, wb)
    paint = tmp_path / "chunk_0" / "paint"
    paint.mkdir(parents=True)
    R = ra.record_bytes(90)
    (paint / "relate_3.bin").write_bytes(b"\0" * (90 * R))  # a complete-looking file of an earlier run
    child = textwrap.dedent(f"""
        import resource, signal, sys
        sys.path.insert(0, {ROOT!r})
        import relate_amd as ra
        # a complete run elsewhere first: the HIP runtime writes files of its own while it loads the kernels
        ra.paint_chunk({str(warm)!r}, 0, 0.001, 1.0, device=0)
        signal.signal(signal.SIGXFSZ, signal.SIG_IGN)
        resource.setrlimit(resource.RLIMIT_FSIZE, ({90 * R // 2}, {90 * R // 2}))
        try:
            ra.paint_chunk({str(tmp_path)!r}, 0, 0.001, 1.0, device=0)
        except ra.RelateError as e:
            print("code", e.code, e)
            sys.exit(7)
        sys.exit(0)
    """)
    p = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300)
    assert p.returncode == 7, p.stdout + p.stderr
    assert "code -2" in p.stdout  # RELATE_EIO
    assert sorted(os.listdir(paint)) == [], "a failed run must leave neither relate_<w>.bin nor .part files"
    # the same call without the limit writes all five files, and no .part file stays behind
    ra.paint_chunk(tmp_path, 0, 0.001, 1.0, device=0)
    assert sorted(os.listdir(paint)) == [f"relate_{w}.bin" for w in range(5)]
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb)
    for w in range(5):
        raw = np.fromfile(paint / f"relate_{w}.bin", dtype=np.uint8)
        assert raw.size == 90 * R
        assert np.array_equal(ol.decode_records(raw.reshape(1, 90, R))["a_snp"], ol.decode_records(want[w:w + 1])["a_snp"])


def test_unwritable_window_file_is_reported_and_nothing_is_left(tmp_path):
    import relate_amd as ra
    seq, r, wb = sy.make_chunk(N=40, L=600, W=3, seed=42, r_mode="flat")
    ra.write_chunk(tmp_path, 0, seq, r, wb)
    paint = tmp_path / "chunk_0" / "paint"
    paint.mkdir(parents=True)
    (paint / "relate_1.bin.part").mkdir()  # cannot be opened for writing
    with pytest.raises(ra.RelateError) as e:
        ra.paint_chunk(tmp_path, 0, 0.001, 1.0, device=0)
    assert e.value.code == -2
    assert sorted(os.listdir(paint)) == ["relate_1.bin.part"]


def test_paint_to_files_of_a_shard_and_staging_release(tmp_path):
    """relate_plan_paint_to_files: two shards of one chunk written through the library's copy-back into the same
    files, the second first; then the page-locked staging block is released."""
    import relate_amd as ra
    from test_gpu_parity import compare_records
    N, L, W = 300, 1500, 4
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=43, r_mode="skew")
    plan = ra.Plan(0)
    plan.load_host(seq, r, wb)
    nd = plan.derived_counts()
    assert np.array_equal(nd, ol.count_derived(seq))
    from relate_amd.dist import shard_by_work
    b = shard_by_work(nd, 2)
    R = ra.record_bytes(N)
    fds = [os.open(tmp_path / f"relate_{w}.bin", os.O_WRONLY | os.O_CREAT, 0o666) for w in range(W)]
    try:
        plan.paint_to_files(0.001, 0.001, int(b[1]), N, fds)
        plan.paint_to_files(0.001, 0.001, 0, int(b[1]), fds)
    finally:
        for fd in fds:
            os.close(fd)
    plan.close()
    ra.lib().relate_release_host_staging()
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb)
    for w in range(W):
        raw = np.fromfile(tmp_path / f"relate_{w}.bin", dtype=np.uint8)
        assert raw.size == N * R
        assert compare_records(raw.reshape(1, N, R), want[w:w + 1], N, f"shards, window {w}") == 0.0


def test_several_plans_share_one_chunk(tmp_path, monkeypatch):
    """relate_paint_chunk(device < 0), the in-process multi-device host: one plan and one host thread per entry of the
    device list, the packed matrix handed from the first plan to the others device to device, target ranges of equal
    work, every thread writing its records at k*R of the same files.  With one GPU the list names it several times
    (RELATE_DEVICES=0,0,0): everything but the second card runs.  Files byte-identical to the one-device run's."""
    import relate_amd as ra
    N, W = 333, 4
    seq, r, wb = sy.make_chunk(N=N, L=1800, W=W, seed=44, r_mode="skew")
    one, many = tmp_path / "one", tmp_path / "many"
    for d in (one, many):
        d.mkdir()
        ra.write_chunk(d, 0, seq, r, wb)
    ra.paint_chunk(one, 0, 0.001, 1.0, device=0)
    monkeypatch.setenv("RELATE_DEVICES", "0,0,0")
    ra.paint_chunk(many, 0, 0.001, 1.0, device=-1)
    R = ra.record_bytes(N)
    assert sorted(os.listdir(many / "chunk_0" / "paint")) == [f"relate_{w}.bin" for w in range(W)]
    for w in range(W):
        a = np.fromfile(one / "chunk_0" / "paint" / f"relate_{w}.bin", dtype=np.uint8)
        b = np.fromfile(many / "chunk_0" / "paint" / f"relate_{w}.bin", dtype=np.uint8)
        assert a.size == N * R and np.array_equal(a, b), f"window {w}"
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb)
    got = np.stack([np.fromfile(many / "chunk_0" / "paint" / f"relate_{w}.bin", dtype=np.uint8).reshape(N, R) for w in range(W)])
    assert np.array_equal(ol.decode_records(got)["a_snp"], ol.decode_records(want)["a_snp"])
    # a list that names a device the machine does not have is refused (the files of the run above stay as they are)
    monkeypatch.setenv("RELATE_DEVICES", "0,99")
    with pytest.raises(ra.RelateError) as e:
        ra.paint_chunk(many, 0, 0.001, 1.0, device=-1)
    assert e.value.code == -1 and "RELATE_DEVICES" in str(e.value)


def test_batches_of_targets_through_two_record_buffers(monkeypatch):
    """relate_paint_stones with the targets in many batches (RELATE_BATCH_BYTES shrinks them): each batch's records are
    copied back while the next batch is painted into the other buffer; every record against the oracle."""
    import relate_amd as ra
    from test_gpu_parity import compare_records
    N, W = 200, 6
    seq, r, wb = sy.make_chunk(N=N, L=3000, W=W, seed=45, r_mode="flat")
    R = ra.record_bytes(N)
    monkeypatch.setenv("RELATE_BATCH_BYTES", str(23 * (W * R + 28 * 400)))  # about twenty targets per batch
    got = ra.paint_stones(seq, r, 0.001, 0.001, wb, device=0)
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb)
    assert compare_records(got, want, N, "nine batches") == 0.0
    got = ra.paint_stones(seq, r, 0.001, 0.01, wb, device=0, k_begin=37, k_end=171)
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.01, wb, 37, 171)
    assert compare_records(got, want, N, "a range in batches, two thetas") == 0.0
