"""Painting one chunk on several GPUs of one node: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), targets sharded in contiguous ranges, no collective on the data path.

    rank 0 : reads chunk_<c>.hap / .r, uploads, bit-packs on its GPU
    all    : ONE broadcast of the packed matrix (+ r) from rank 0             <- the only exchange
    rank g : paints its range of targets -- ranges of equal numbers of derived sites, which every rank works
             out for itself from the packed matrix it holds -- and writes the records at their fixed place
             k*(56+8N) of relate_<w>.bin.part (pwrite), so that the files are byte-identical to the ones a
             single process writes (pipeline/Paint.cpp:78-82 appends k = 0..N-1)
    rank 0 : renames the .part files once every rank has reported success, and removes them otherwise:
             a relate_<w>.bin exists only complete (consumers trust its size, src/anc_builder.cpp:65-69)

The painting itself is delegated to a backend object (GpuBackend below drives the HIP library); the
orchestration is independent of it, which is how the multi-process logic is tested on CPU with gloo."""
import ctypes as C
import os

import numpy as np

from ._lib import i32p, lib, record_bytes


def shard_range(N, rank, world):
    """Contiguous target range of `rank` by count; the union over ranks is [0,N) and sizes differ by at most one."""
    return (N * rank) // world, (N * (rank + 1)) // world


def shard_by_work(nd, world):
    """Bounds (world+1 ints) of contiguous target ranges with (nearly) equal sums of nd, the number of derived
    sites of a target, to which the cost of painting it is proportional.  Host-only library routine."""
    nd = np.ascontiguousarray(nd, dtype=np.int32)
    bounds = np.zeros(world + 1, dtype=np.int32)
    rc = lib().relate_shard_by_work(len(nd), nd.ctypes.data_as(i32p), world, bounds.ctypes.data_as(i32p))
    if rc != 0:
        raise ValueError("relate_shard_by_work: bad argument")
    return bounds


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
    """Drives librelate_gpu.so on the rank's GPU; tensors are torch tensors on that device.  Every library call
    is given torch's current stream, so the packing, the broadcast (which torch orders on that stream) and the
    kernels that read the broadcast matrix follow one another without host-side waits."""

    def __init__(self, device):
        import torch
        from .paint import Plan
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.plan = Plan(device)
        self.keep = []

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def packed_tensor(self, N, L):
        return self.torch.empty(lib().relate_packed_bytes(N, L), dtype=self.torch.uint8, device=self.device)

    def r_tensor(self, L):
        return self.torch.empty(L, dtype=self.torch.float64, device=self.device)

    def flag_tensor(self, value):
        return self.torch.tensor([value], dtype=self.torch.int32, device=self.device)

    def pack(self, seq, r, wb, packed, d_r):
        """rank 0: upload the byte matrix and pack it into `packed`."""
        t = self.torch
        L, N = seq.shape
        d_seq = t.from_numpy(np.ascontiguousarray(seq)).to(self.device)
        d_r.copy_(t.from_numpy(np.ascontiguousarray(r)))
        self.plan.load_device_into(N, L, d_seq.data_ptr(), d_r.data_ptr(), wb, packed.data_ptr(), packed.numel(),
                                   self._stream())
        t.cuda.current_stream(self.device).synchronize()  # d_seq is dropped on return
        self.keep = [packed, d_r]

    def adopt(self, N, L, wb, packed, d_r):
        self.plan.load_packed(N, L, packed.data_ptr(), packed.numel(), d_r.data_ptr(), wb, self._stream())
        self.keep = [packed, d_r]

    def derived_counts(self):
        self.torch.cuda.current_stream(self.device).synchronize()  # the library counts on a stream of its own
        return self.plan.derived_counts()

    def paint(self, theta_p, theta_d, k0, k1):
        """Records of targets [k0,k1) as a (W, k1-k0, R) uint8 array on the host."""
        t = self.torch
        st = self._stream()
        self.plan.prepare(theta_p, theta_d, k0, k1, st)
        out = t.empty(self.plan.out_bytes(), dtype=t.uint8, device=self.device)
        self.plan.paint(out.data_ptr(), st)
        t.cuda.current_stream(self.device).synchronize()
        return out.cpu().numpy().reshape(self.plan.W, k1 - k0, record_bytes(self.plan.N))

    def timer(self):
        """Stage times of a step from events on torch's current stream (collectives are ordered on it too)."""
        torch, dev = self.torch, self.device

        class _Timer:
            def __init__(self):
                self.ev = [torch.cuda.Event(enable_timing=True)]
                self.names = []
                self.ev[0].record(torch.cuda.current_stream(dev))

            def mark(self, name):
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream(dev))
                self.ev.append(e)
                self.names.append(name)

            def result(self):
                self.ev[-1].synchronize()
                return {n: self.ev[i].elapsed_time(self.ev[i + 1]) for i, n in enumerate(self.names)}

        return _Timer()

    def pack_device(self, d_seq, d_r_in, N, L, wb, packed, d_r):
        """rank 0: pack the byte matrix that already lies in HBM into `packed`; r into the tensor that is broadcast."""
        if d_r.data_ptr() != d_r_in.data_ptr():
            d_r.copy_(d_r_in)
        self.plan.load_device_into(N, L, d_seq.data_ptr(), d_r.data_ptr(), wb, packed.data_ptr(), packed.numel(),
                                   self._stream())
        self.keep = [packed, d_r]

    def paint_resident(self, theta_p, theta_d, k0, k1, out=None):
        """Paints [k0,k1) into a device tensor (reused when large enough) and leaves the records there."""
        t = self.torch
        st = self._stream()
        self.plan.prepare(theta_p, theta_d, k0, k1, st)
        need = self.plan.out_bytes()
        if out is None or out.numel() < need:
            out = None
            out = t.empty(need, dtype=t.uint8, device=self.device)
        self.plan.paint(out.data_ptr(), st)
        return out

    def paint_to_files(self, theta_p, theta_d, k0, k1, fds):
        """Paints [k0,k1) and writes the records at k*R of the open files `fds` (one per window): batching,
        page-locked copy-back and the writing threads are the library's (relate_plan_paint_to_files)."""
        self.torch.cuda.current_stream(self.device).synchronize()  # the library paints on a stream of its own
        self.plan.paint_to_files(theta_p, theta_d, k0, k1, fds)


class ShardedChunk:
    """One chunk painted by every rank of the default process group with its records left in each rank's HBM:
    what the multi-GPU benchmark times.  Per step: rank 0 packs the haplotype bytes that lie in its HBM, ONE
    broadcast of the packed matrix and one of r (the only exchange), every rank works out the same contiguous
    ranges of equal work from the packed matrix it now holds and paints its own.

    `backend` as in paint_chunk_distributed; `device_seq` / `device_r` are rank 0's device tensors (ignored on the
    other ranks).  Timings of the last step, milliseconds of this rank: .last = {pack, broadcast, shard, paint}."""

    def __init__(self, backend, N, L, wb, theta_painter, theta_data):
        import torch.distributed as dist
        self.dist = dist
        self.be = backend
        self.N, self.L, self.wb = N, L, np.ascontiguousarray(wb, dtype=np.int32)
        self.tp, self.td = float(theta_painter), float(theta_data)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.packed = backend.packed_tensor(N, L)
        self.d_r = backend.r_tensor(L)
        self.range = (0, 0)
        self.out = None
        self.sum_derived = 0
        self.last = {}

    def step(self, device_seq=None, device_r=None):
        be, dist = self.be, self.dist
        t = be.timer()
        if self.rank == 0:
            be.pack_device(device_seq, device_r, self.N, self.L, self.wb, self.packed, self.d_r)
        t.mark("pack")
        if self.world > 1:
            dist.broadcast(self.packed, src=0)
            dist.broadcast(self.d_r, src=0)
        t.mark("broadcast")
        if self.rank != 0:
            be.adopt(self.N, self.L, self.wb, self.packed, self.d_r)
        nd = be.derived_counts()
        bounds = shard_by_work(nd, self.world)
        k0, k1 = int(bounds[self.rank]), int(bounds[self.rank + 1])
        self.range = (k0, k1)
        t.mark("shard")
        self.sum_derived = int(nd.sum())
        if k1 > k0:  # more ranks than targets with work leaves some ranges empty
            self.out = be.paint_resident(self.tp, self.td, k0, k1, self.out)
        t.mark("paint")
        self.last = t.result()
        return self.sum_derived


def paint_chunk_distributed(output_dir, chunk_index, theta=None, rho=None, backend=None, balance=True):
    """Paint stage of one chunk over all ranks of the default process group (initialise it first).

    Returns the (k0, k1) range this rank painted.  Raises on every rank if any rank failed; no relate_<w>.bin of
    the chunk exists then."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    N, L, wb = read_parameters(output_dir, chunk_index)
    W = len(wb) - 1
    theta_v = 0.001 if theta is None else float(theta)  # Data's default, src/data.cpp:97
    paint_dir = os.path.join(str(output_dir), f"chunk_{chunk_index}", "paint")
    finals = [os.path.join(paint_dir, f"relate_{w}.bin") for w in range(W)]
    parts = [f + ".part" for f in finals]
    # Every step that can fail on one rank alone sits between collectives that all ranks reach: a rank that fails
    # keeps taking part in them (with whatever its buffers hold) and reports through the all_reduce at the end, so
    # no rank is left waiting in a collective and every rank raises.
    ok, err = 1, None

    def attempt(step):
        nonlocal ok, err
        if not ok:
            return None
        try:
            return step()
        except Exception as e:  # reported to every rank below
            ok, err = 0, e
            return None

    bufs = attempt(lambda: (backend.packed_tensor(N, L), backend.r_tensor(L)))
    if bufs is None:  # without its buffers a rank cannot take part in the broadcast: fail before any collective
        raise RuntimeError(f"painting of chunk {chunk_index} failed on rank {rank} before the hand-off: {err}")
    packed, d_r = bufs
    if rank == 0:
        def load():
            seq, r = read_chunk(output_dir, chunk_index)
            if rho is not None:
                r = r * float(rho)
            backend.pack(seq, r, wb, packed, d_r)
        attempt(load)
    dist.broadcast(packed, src=0)
    dist.broadcast(d_r, src=0)
    if rank != 0:
        attempt(lambda: backend.adopt(N, L, wb, packed, d_r))
    if rank == 0:
        def create():
            os.makedirs(paint_dir, exist_ok=True)
            for f, p in zip(finals, parts):  # "wb" in pipeline/Paint.cpp:74: nothing of an earlier run survives
                if os.path.exists(f):
                    os.unlink(f)
                with open(p, "wb"):
                    pass
        attempt(create)
    dist.barrier()
    k0, k1 = shard_range(N, rank, world)
    if balance and world > 1:
        b = attempt(lambda: shard_by_work(backend.derived_counts(), world))
        if b is not None:
            k0, k1 = int(b[rank]), int(b[rank + 1])
    fds = []
    try:
        if ok:
            fds = [os.open(p, os.O_WRONLY) for p in parts]
            if k1 > k0:
                backend.paint_to_files(theta_v, theta_v, k0, k1, fds)
    except Exception as e:
        ok, err = 0, e
    finally:
        for fd in fds:
            os.close(fd)
    flag = backend.flag_tensor(ok)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    all_ok = int(flag.item()) == 1
    if rank == 0:
        for f, p in zip(finals, parts):
            if all_ok:
                os.replace(p, f)
            elif os.path.exists(p):
                os.unlink(p)
    dist.barrier()
    if not all_ok:
        raise RuntimeError(f"painting of chunk {chunk_index} failed on rank {rank if err else '(another)'}: {err}")
    return k0, k1
