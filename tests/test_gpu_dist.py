"""The multi-process host on a real GPU: relate_amd.dist.paint_chunk_distributed with the GPU backend in a
process group of one rank over RCCL (the broadcast, the packed hand-off and the sharded writes are the code the
N-rank run executes), and the adoption of a packed matrix by a second plan, which is what ranks > 0 do."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import synth_np as sy
from test_gpu_parity import compare_records

pytestmark = pytest.mark.gpu


def test_one_rank_group_over_rccl(tmp_path):
    import torch
    import torch.distributed as dist
    import relate_amd as ra
    from relate_amd.dist import GpuBackend, paint_chunk_distributed
    seq, r, wb = sy.make_chunk(N=75, L=1100, W=3, seed=71, r_mode="flat")
    ra.write_chunk(tmp_path, 0, seq, r, wb)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29600 + os.getpid() % 300}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        k0, k1 = paint_chunk_distributed(tmp_path, 0, theta=0.002, rho=1.25, backend=GpuBackend(0))
    finally:
        dist.destroy_process_group()
    assert (k0, k1) == (0, 75)
    want = ol.oracle().paint_stones(seq, r * 1.25, 0.002, 0.002, wb)
    R = ra.record_bytes(75)
    for w in range(3):
        raw = np.fromfile(tmp_path / "chunk_0" / "paint" / f"relate_{w}.bin", dtype=np.uint8)
        assert raw.size == 75 * R
        compare_records(raw.reshape(1, 75, R), want[w:w + 1], 75, f"window {w}")
    assert not [f for f in os.listdir(tmp_path / "chunk_0" / "paint") if f.endswith(".part")]


def test_packed_matrix_adopted_by_a_second_plan():
    import torch
    from relate_amd.dist import GpuBackend
    seq, r, wb = sy.make_chunk(N=200, L=800, W=2, seed=72, r_mode="flat")
    a, b = GpuBackend(0), GpuBackend(0)
    packed, d_r = a.packed_tensor(200, 800), a.r_tensor(800)
    a.pack(seq, r, wb, packed, d_r)
    clone, r_clone = packed.clone(), d_r.clone()   # what a broadcast leaves on another rank
    torch.cuda.synchronize()
    b.adopt(200, 800, wb, clone, r_clone)
    got = b.paint(0.001, 0.001, 120, 200)
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb, 120, 200)
    compare_records(got, want, 200, "adopted")
    also = a.paint(0.001, 0.001, 0, 40)
    compare_records(also, ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb, 0, 40), 200, "packer")
