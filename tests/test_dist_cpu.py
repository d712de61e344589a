"""The N>1 path on CPU: two processes over gloo run relate_amd.dist.paint_chunk_distributed with a backend
whose painting is done by the CPU oracle (the GPU is not needed to exercise the orchestration: one
broadcast of the packed matrix from rank 0, target shards of equal work, records written at fixed offsets,
files that appear complete or not at all)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


class OracleBackend:
    """Same interface as relate_amd.dist.GpuBackend; 'packing' is numpy.packbits, painting is the oracle."""

    def __init__(self, fail_on_rank=None, rank=0):
        self.seq = self.r = self.wb = None
        self.fail = fail_on_rank is not None and fail_on_rank == rank

    def packed_tensor(self, N, L):
        return torch.zeros(8 + (N * L + 7) // 8, dtype=torch.uint8)

    def r_tensor(self, L):
        return torch.zeros(L, dtype=torch.float64)

    def flag_tensor(self, value):
        return torch.tensor([value], dtype=torch.int32)

    def pack(self, seq, r, wb, packed, d_r):
        L, N = seq.shape
        bits = np.packbits(seq == ord("1"))
        packed[:8] = torch.from_numpy(np.array([N, L], dtype=np.int32).view(np.uint8))
        packed[8:] = torch.from_numpy(bits)
        d_r.copy_(torch.from_numpy(r))
        self.adopt(N, L, wb, packed, d_r)

    def adopt(self, N, L, wb, packed, d_r):
        n, l = packed[:8].numpy().view(np.int32)
        assert (n, l) == (N, L)
        bits = np.unpackbits(packed[8:].numpy())[: N * L].reshape(L, N)
        self.seq = (bits + ord("0")).astype(np.uint8)
        self.r = d_r.numpy().copy()
        self.wb = np.asarray(wb, dtype=np.int32)

    def derived_counts(self):
        import oracle_lib as ol
        return ol.count_derived(self.seq)

    # what ShardedChunk needs on top of the above
    def timer(self):
        import time

        class _T:
            def __init__(self):
                self.t = [time.perf_counter()]
                self.names = []

            def mark(self, name):
                self.t.append(time.perf_counter())
                self.names.append(name)

            def result(self):
                return {n: 1e3 * (self.t[i + 1] - self.t[i]) for i, n in enumerate(self.names)}

        return _T()

    def pack_device(self, d_seq, d_r_in, N, L, wb, packed, d_r):
        self.pack(d_seq.numpy(), d_r_in.numpy(), wb, packed, d_r)

    def paint_resident(self, theta_p, theta_d, k0, k1, out=None):
        import oracle_lib as ol
        return torch.from_numpy(ol.oracle().paint_stones(self.seq, self.r, theta_p, theta_d, self.wb, k0, k1))

    def paint_to_files(self, theta_p, theta_d, k0, k1, fds):
        import oracle_lib as ol
        R = ol.record_size(self.seq.shape[1])
        half = (k0 + k1) // 2
        for a, b in ((k0, half), (half, k1)):
            if b <= a:
                continue
            rec = ol.oracle().paint_stones(self.seq, self.r, theta_p, theta_d, self.wb, a, b)
            for w, fd in enumerate(fds):
                os.pwrite(fd, rec[w].tobytes(), a * R)
            if self.fail:
                raise RuntimeError("injected failure after the first batch")


def _worker(rank, world, port, out_dir, result_dir, fail_on_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from relate_amd.dist import paint_chunk_distributed
        be = OracleBackend(fail_on_rank, rank)
        try:
            k0, k1 = paint_chunk_distributed(out_dir, 0, theta=0.002, rho=1.5, backend=be)
            np.save(os.path.join(result_dir, f"range_{rank}.npy"), np.array([k0, k1]))
        except RuntimeError as e:
            with open(os.path.join(result_dir, f"error_{rank}.txt"), "w") as f:
                f.write(str(e))
    finally:
        dist.destroy_process_group()


def _chunk(tmp_path):
    import relate_amd as ra
    import synth_np as sy
    seq, r, wb = sy.make_chunk(N=45, L=900, W=4, seed=9, r_mode="flat")
    # make the work uneven: the last third of the haplotypes carries the derived allele at many more sites
    rng = np.random.default_rng(1)
    extra = rng.random((900, 15)) < 0.3
    seq[:, 30:][extra] = ord("1")
    out = tmp_path / "out"
    out.mkdir()
    ra.write_chunk(out, 0, seq, r, wb)
    res = tmp_path / "res"
    res.mkdir()
    return seq, r, wb, out, res


@pytest.mark.timeout(300)
def test_two_ranks_write_the_same_files_as_one_process(tmp_path):
    import oracle_lib as ol
    import relate_amd as ra
    from relate_amd.dist import shard_by_work
    seq, r, wb, out, res = _chunk(tmp_path)
    (out / "chunk_0" / "paint").mkdir(parents=True)
    (out / "chunk_0" / "paint" / "relate_1.bin").write_bytes(b"stale")  # of an earlier run: must not survive as is
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(out), str(res), None), nprocs=2, join=True)
    ranges = [tuple(int(v) for v in np.load(res / f"range_{i}.npy")) for i in range(2)]
    nd = ol.count_derived(seq)
    b = shard_by_work(nd, 2)
    assert ranges == [(0, int(b[1])), (int(b[1]), 45)]
    assert b[1] > 23, "the heavier targets sit at the end: the first shard must hold more targets than half"
    want = ol.oracle().paint_stones(seq, r * 1.5, 0.002, 0.002, wb)
    R = ra.record_bytes(45)
    for w in range(4):
        raw = np.fromfile(out / "chunk_0" / "paint" / f"relate_{w}.bin", dtype=np.uint8)
        assert raw.size == 45 * R
        assert np.array_equal(raw.reshape(45, R), want[w])
    assert not list((out / "chunk_0" / "paint").glob("*.part"))


@pytest.mark.timeout(300)
def test_failure_on_one_rank_leaves_no_window_file(tmp_path):
    """A rank that fails after part of its shard has been written: every rank raises, and neither a complete-looking
    relate_<w>.bin (consumers trust the size, src/anc_builder.cpp:65-69) nor a .part file is left."""
    seq, r, wb, out, res = _chunk(tmp_path)
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(out), str(res), 1), nprocs=2, join=True)
    assert (res / "error_0.txt").exists() and (res / "error_1.txt").exists()
    assert "injected failure" in (res / "error_1.txt").read_text()
    assert list((out / "chunk_0" / "paint").iterdir()) == []


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_of_equal_work(world):
    from relate_amd.dist import shard_by_work
    rng = np.random.default_rng(world)
    nd = np.concatenate([rng.integers(2, 50, 300), rng.integers(500, 900, 100)]).astype(np.int32)
    b = shard_by_work(nd, world)
    assert b[0] == 0 and b[-1] == len(nd) and np.all(np.diff(b) >= 1)
    sums = np.array([nd[b[i]:b[i + 1]].sum() for i in range(world)])
    assert sums.max() - sums.min() <= 2 * nd.max()
    # more parts than targets: the first ranges take one target each, the rest stay empty-free at the end
    b = shard_by_work(np.array([5, 7, 9], dtype=np.int32), 3)
    assert list(b) == [0, 1, 2, 3]


def _resident_worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synth_np as sy
        from relate_amd.dist import ShardedChunk
        seq, r, wb = sy.make_chunk(N=40, L=700, W=3, seed=19, r_mode="flat")
        be = OracleBackend()
        sc = ShardedChunk(be, 40, 700, wb, 0.001, 0.001)
        d_seq = torch.from_numpy(seq) if rank == 0 else None   # only rank 0 holds the inputs
        d_r = torch.from_numpy(r) if rank == 0 else None
        for _ in range(2):
            total = sc.step(d_seq, d_r)
        np.save(os.path.join(result_dir, f"rec_{rank}.npy"), sc.out.numpy())
        np.save(os.path.join(result_dir, f"meta_{rank}.npy"), np.array([sc.range[0], sc.range[1], total]))
        assert set(sc.last) == {"pack", "broadcast", "shard", "paint"}
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_resident_sharded_step_covers_the_chunk(tmp_path):
    """ShardedChunk (what bench.py --gpus N times): the ranks' resident records, side by side, are the chunk's."""
    import oracle_lib as ol
    import synth_np as sy
    seq, r, wb = sy.make_chunk(N=40, L=700, W=3, seed=19, r_mode="flat")
    port = 33500 + os.getpid() % 2000
    mp.spawn(_resident_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    meta = [np.load(tmp_path / f"meta_{i}.npy") for i in range(3)]
    assert meta[0][0] == 0 and meta[2][1] == 40 and meta[0][1] == meta[1][0] and meta[1][1] == meta[2][0]
    assert all(int(m[2]) == int(ol.count_derived(seq).sum()) for m in meta)
    got = np.concatenate([np.load(tmp_path / f"rec_{i}.npy") for i in range(3)], axis=1)
    assert np.array_equal(got, ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb))
