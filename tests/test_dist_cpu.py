"""The N>1 path on CPU: two processes over gloo run relate_amd.dist.paint_chunk_distributed with a backend
whose painting is done by the CPU oracle (the GPU is not needed to exercise the orchestration: one
broadcast of the packed matrix from rank 0, contiguous target shards, records written at fixed offsets)."""
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

    def __init__(self):
        self.seq = self.r = self.wb = None

    def packed_tensor(self, N, L):
        return torch.zeros(8 + (N * L + 7) // 8, dtype=torch.uint8)

    def r_tensor(self, L):
        return torch.zeros(L, dtype=torch.float64)

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

    def paint(self, theta_p, theta_d, k0, k1):
        import oracle_lib as ol
        return ol.oracle().paint_stones(self.seq, self.r, theta_p, theta_d, self.wb, k0, k1)


def _worker(rank, world, port, out_dir, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from relate_amd.dist import paint_chunk_distributed
        be = OracleBackend()
        k0, k1 = paint_chunk_distributed(out_dir, 0, theta=0.002, rho=1.5, backend=be, batch_bytes=40000)
        # only rank 0 read the files: the others got the chunk through the broadcast
        np.save(os.path.join(result_dir, f"range_{rank}.npy"), np.array([k0, k1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_write_the_same_files_as_one_process(tmp_path):
    import oracle_lib as ol
    import relate_amd as ra
    import synth_np as sy
    seq, r, wb = sy.make_chunk(N=45, L=900, W=4, seed=9, r_mode="flat")
    out = tmp_path / "out"
    out.mkdir()
    ra.write_chunk(out, 0, seq, r, wb)
    res = tmp_path / "res"
    res.mkdir()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(out), str(res)), nprocs=2, join=True)
    ranges = [tuple(np.load(res / f"range_{i}.npy")) for i in range(2)]
    assert ranges == [(0, 22), (22, 45)]
    want = ol.oracle().paint_stones(seq, r * 1.5, 0.002, 0.002, wb)
    R = ra.record_bytes(45)
    for w in range(4):
        raw = np.fromfile(out / "chunk_0" / "paint" / f"relate_{w}.bin", dtype=np.uint8)
        assert raw.size == 45 * R
        assert np.array_equal(raw.reshape(45, R), want[w])
