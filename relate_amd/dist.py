"""Painting one chunk on several GPUs of one node: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), targets sharded in contiguous ranges, no collective on the data path.

    rank 0 : reads chunk_<c>.hap / .r, uploads, bit-packs on its GPU
    all    : ONE broadcast of the packed matrix (+ r) from rank 0             <- the only exchange
    rank g : paints targets [N*g/G, N*(g+1)/G) in batches, copies the records back over its own PCIe link and
             writes them at their fixed place k*(56+8N) of relate_<w>.bin (pwrite), so that the files are
             byte-identical to the ones a single process writes (pipeline/Paint.cpp:78-82 appends k = 0..N-1)

The painting itself is delegated to a backend object (GpuBackend below drives the HIP library); the
orchestration is independent of it, which is how the multi-process logic is tested on CPU with gloo."""
import os

import numpy as np

from ._lib import record_bytes


def shard_range(N, rank, world):
    """Contiguous target range of `rank`; the union over ranks is [0,N) and sizes differ by at most one."""
    return (N * rank) // world, (N * (rank + 1)) // world


def read_parameters(output_dir, chunk_index):
    """parameters_c<c>.bin: int32 N, L, W+1, wb[W+1]  (pipeline/Paint.cpp:21-29)."""
    raw = np.fromfile(os.path.join(str(output_dir), f"parameters_c{chunk_index}.bin"), dtype=np.int32)
    N, L, nw = int(raw[0]), int(raw[1]), int(raw[2])
    return N, L, raw[3:3 + nw].copy()


def read_chunk(output_dir, chunk_index):
    """chunk_<c>.hap (uint64 L, N, then L*N chars) and chunk_<c>.r (uint32 L, then doubles)."""
    base = os.path.join(str(output_dir), f"chunk_{chunk_index}")
    with open(base + ".hap", "rb") as f:
        L, N = (int(v) for v in np.fromfile(f, dtype=np.uint64, count=2))
        seq = np.fromfile(f, dtype=np.uint8, count=L * N).reshape(L, N)
    with open(base + ".r", "rb") as f:
        n = int(np.fromfile(f, dtype=np.uint32, count=1)[0])
        r = np.fromfile(f, dtype=np.float64, count=n)
    return seq, r


class GpuBackend:
    """Drives librelate_gpu.so on the rank's GPU; tensors are torch tensors on that device."""

    def __init__(self, device):
        import torch
        from .paint import Plan
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.plan = Plan(device)
        self.keep = []

    def packed_tensor(self, N, L):
        from ._lib import lib
        return self.torch.empty(lib().relate_packed_bytes(N, L), dtype=self.torch.uint8, device=self.device)

    def r_tensor(self, L):
        return self.torch.empty(L, dtype=self.torch.float64, device=self.device)

    def pack(self, seq, r, wb, packed, d_r):
        """rank 0: upload the byte matrix and pack it into `packed`."""
        t = self.torch
        L, N = seq.shape
        d_seq = t.from_numpy(np.ascontiguousarray(seq)).to(self.device)
        d_r.copy_(t.from_numpy(np.ascontiguousarray(r)))
        st = t.cuda.current_stream(self.device).cuda_stream
        self.plan.load_device_into(N, L, d_seq.data_ptr(), d_r.data_ptr(), wb, packed.data_ptr(), packed.numel(), st)
        t.cuda.synchronize(self.device)
        self.keep = [packed, d_r]

    def adopt(self, N, L, wb, packed, d_r):
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        self.plan.load_packed(N, L, packed.data_ptr(), packed.numel(), d_r.data_ptr(), wb, st)
        self.keep = [packed, d_r]

    def paint(self, theta_p, theta_d, k0, k1):
        """Records of targets [k0,k1) as a (W, k1-k0, R) uint8 array on the host."""
        t = self.torch
        st = t.cuda.current_stream(self.device).cuda_stream
        self.plan.prepare(theta_p, theta_d, k0, k1, st)
        out = t.empty(self.plan.out_bytes(), dtype=t.uint8, device=self.device)
        self.plan.paint(out.data_ptr(), st)
        t.cuda.synchronize(self.device)
        return out.cpu().numpy().reshape(self.plan.W, k1 - k0, record_bytes(self.plan.N))


def paint_chunk_distributed(output_dir, chunk_index, theta=None, rho=None, backend=None, batch_bytes=2 << 30):
    """Paint stage of one chunk over all ranks of the default process group (initialise it first).

    Returns the (k0, k1) range this rank painted."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    N, L, wb = read_parameters(output_dir, chunk_index)
    W = len(wb) - 1
    R = record_bytes(N)
    theta_v = 0.001 if theta is None else float(theta)  # Data's default, src/data.cpp:97
    packed = backend.packed_tensor(N, L)
    d_r = backend.r_tensor(L)
    if rank == 0:
        seq, r = read_chunk(output_dir, chunk_index)
        if rho is not None:
            r = r * float(rho)
        backend.pack(seq, r, wb, packed, d_r)
    dist.broadcast(packed, src=0)
    dist.broadcast(d_r, src=0)
    if rank != 0:
        backend.adopt(N, L, wb, packed, d_r)
    paint_dir = os.path.join(str(output_dir), f"chunk_{chunk_index}", "paint")
    if rank == 0:
        os.makedirs(paint_dir, exist_ok=True)
        for w in range(W):  # "wb" in pipeline/Paint.cpp:74: start from empty files
            with open(os.path.join(paint_dir, f"relate_{w}.bin"), "wb"):
                pass
    dist.barrier()
    k0, k1 = shard_range(N, rank, world)
    fds = [os.open(os.path.join(paint_dir, f"relate_{w}.bin"), os.O_WRONLY) for w in range(W)]
    try:
        kb = max(1, min(N, int(batch_bytes // max(1, W * R))))
        for a in range(k0, k1, kb):
            b = min(k1, a + kb)
            rec = backend.paint(theta_v, theta_v, a, b)
            for w in range(W):
                os.pwrite(fds[w], rec[w].tobytes(), a * R)
    finally:
        for fd in fds:
            os.close(fd)
    dist.barrier()
    return k0, k1
