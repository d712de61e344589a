#!/usr/bin/env python3
"""Benchmark of the Paint hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]

A step is one complete Paint of one chunk whose inputs (the '0'/'1' haplotype matrix and r) already lie in
HBM: bit-packing + transpose, per-target precompute, forward/backward recursion of all N targets, records
written to HBM in their file layout.

With N > 1 (launched by torch.distributed.run, one rank per GPU) the job still paints ONE chunk per step: the
inputs lie in rank 0's HBM, rank 0 packs, the packed matrix (and r) is broadcast once over RCCL, every rank
paints a contiguous range of targets holding 1/N of the derived sites and keeps its records in its own HBM.
Total work is fixed ("strong" scaling); there is no reduction.  --independent-chunks gives every rank a chunk
of its own instead (how the reference itself uses several processes, one per --chunk_index,
pipeline/Relate.cpp:217-229; "weak").

Prints ONE JSON line (rank 0).  value = N(N-1)L cell-updates per second of the whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD_MEMORY_GB = {"c4": 40.0}  # --memory of MakeChunks for the window rule (default 5, src/data.cpp:145)
WORKLOADS = {
    # name: (N, L, description)
    "c2": (2000, 50000, "Paint-only single chunk: 2000 haplotypes x 50k SNPs synthetic"),
    "c3": (5008, 100000, "1000G-scale: 5008 haplotypes, 100k-SNP synthetic chunk"),
    "c4": (20000, 100000, "20k haplotypes x 100k SNPs (windows by the memory rule with --memory 40), painted in batches of targets"),
    "c5": (10000, 50000, "rare-variant-dense: 10k haplotypes x 50k SNPs, skewed recombination map"),
    "tiny": (256, 4000, "tiny functional check"),
}
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X vector FP64 (FMA = 2 flop), spec
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, spec


def _cpu_lib():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    if ol.ref_available():
        return ol.ref(), "reference"
    return ol.oracle(), "port"


def cpu_baseline(seq, r, wb, theta, budget_s=12.0, all_cores_budget_s=15.0, o3_budget_s=8.0):
    """The reference's own painting code (oracle/_ref, where it was built) or its C restatement (oracle/), timed
    on the host: on one core -- the reference's execution model, a single-threaded loop over the targets
    (pipeline/Paint.cpp:78-82) -- on all cores of this process's CPU share with the targets split over threads,
    and on one core again with the -O3 / AVX2 build of the same sources.  All on a sample of targets of the same
    chunk: blocks of consecutive targets spread over the panel; the chunk is handed over once (a handle)."""
    import concurrent.futures as cf
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    lib, kind = _cpu_lib()
    L, N = seq.shape
    chunk = lib.open_chunk(seq, r, theta)
    t0 = time.perf_counter()
    first = chunk.paint_stones(theta, wb, 0, 2)
    per_target = (time.perf_counter() - t0) / 2
    block = 4  # consecutive targets per call
    nblocks = int(max(1, min(64, budget_s / max(per_target * block, 1e-6))))
    starts = np.linspace(0, N - block, nblocks).astype(int)
    t0 = time.perf_counter()
    for k in starts:
        chunk.paint_stones(theta, wb, int(k), int(k) + block)
    dt = time.perf_counter() - t0
    ks = nblocks * block
    one = {"value": ks * (N - 1) * L / dt, "unit": "cell-updates/s", "cores": 1, "kind": kind, "build": "-O2",
           "sample": f"{ks} of {N} targets of the same chunk ({nblocks} blocks of {block}), {dt:.1f} s on one core"}
    # all cores: the affinity mask is what this process may use (the machine may have many more)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores) or cores)))
    per_thread = int(max(1, min(16, all_cores_budget_s / max(dt / nblocks, 1e-6))))
    starts_all = np.linspace(0, N - block, cores * per_thread).astype(int)

    def work(k):
        chunk.paint_stones(theta, wb, int(k), int(k) + block)  # ctypes releases the GIL; the chunk is only read

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, starts_all))
    dta = time.perf_counter() - t0
    ka = len(starts_all) * block
    allc = {"value": ka * (N - 1) * L / dta, "unit": "cell-updates/s", "cores": cores, "kind": kind, "build": "-O2",
            "sample": f"{ka} of {N} targets ({len(starts_all)} blocks of {block} over {cores} threads), {dta:.1f} s"}
    chunk.close()
    # the same sources built -O3 -march=x86-64-v3 -ffp-contract=off (same bytes out, checked on the first targets)
    o3 = None
    fast = ol.fast_build(kind)
    if fast is not None:
        try:
            fchunk = fast.open_chunk(seq, r, theta)
            same = bool(np.array_equal(fchunk.paint_stones(theta, wb, 0, 2), first))
            nb3 = int(max(1, min(nblocks, o3_budget_s / max(dt / nblocks, 1e-6))))
            starts3 = np.linspace(0, N - block, nb3).astype(int)
            t0 = time.perf_counter()
            for k in starts3:
                fchunk.paint_stones(theta, wb, int(k), int(k) + block)
            dt3 = time.perf_counter() - t0
            fchunk.close()
            o3 = {"value": nb3 * block * (N - 1) * L / dt3, "unit": "cell-updates/s", "cores": 1, "kind": kind,
                  "build": "-O3 -march=x86-64-v3 -ffp-contract=off", "same_bytes_as_O2": same,
                  "sample": f"{nb3 * block} of {N} targets ({nb3} blocks of {block}), {dt3:.1f} s on one core"}
        except Exception as e:  # an unsupported instruction set on this host, for instance
            o3 = {"skipped": repr(e)}
    return one, allc, o3


def end_to_end(ra, seq, r, wb, theta, theta_data):
    """PCIe-inclusive time of the host-buffer entry point on the whole chunk: haplotype bytes up, packing,
    painting, every record back into pageable host memory.  Never part of `value`."""
    L, N = seq.shape
    W = len(wb) - 1
    need = W * N * ra.record_bytes(N)
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        avail = 0
    if need > 0.5 * avail:
        return {"skipped": f"{need / 2**30:.1f} GiB of records do not fit comfortably in host memory"}
    ra.paint_stones(seq, r, theta, theta_data, wb, 0, min(N, 64))  # pins the staging block, loads the code
    t0 = time.perf_counter()
    out = ra.paint_stones(seq, r, theta, theta_data, wb)
    dt = time.perf_counter() - t0
    nbytes = out.nbytes
    del out
    ra.lib().relate_release_host_staging()
    return {"relate_paint_stones_ms": dt * 1e3, "record_gib": nbytes / 2**30,
            "cell_updates_per_s": float(N) * (N - 1) * L / dt,
            "note": "host buffers in, records back in pageable host memory (allocation of the result included)"}


def repaint_leg(plan, torch, d_out, w, N, R, stream, reps=3):
    """Re-painting of one window from the stones the timed steps left in HBM.  Per element of the topology matrix
    the two launches write the product (4 B) and write and read back every B-th alpha row (16 / B bytes, B =
    checkpoint_rows of the launch geometry; the reference's scheme, every row kept, is B = 1: 20 B)."""
    rows, pitch, _ = plan.repaint_layout(w)
    geo = plan.repaint_geometry()
    top = torch.empty(rows * pitch, dtype=torch.float32, device="cuda")
    ls = torch.empty(rows, dtype=torch.float32, device="cuda")
    ms = []
    for _ in range(reps):
        plan.repaint(w, d_out.data_ptr() + w * N * R, top.data_ptr(), ls.data_ptr(), stream)
        ms.append(plan.timings()["paint_ms"])
    t = min(ms) * 1e-3
    per_element = 4.0 + 16.0 / geo["checkpoint_rows"]
    b_alg = per_element * rows * N
    out = {"window": w, "rows": rows, "pitch": pitch, "geometry": geo, "ms": min(ms), "elements_per_s": rows * N / t,
           "bytes_per_element": per_element, "output_gb_per_s": 4.0 * rows * N / t / 1e9,
           "roofline": {"bound": "hbm", "achieved": b_alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": b_alg / t / 1e9 / HBM_PEAK_GBS, "traffic": None}}
    # distance matrix of one SNP from that topology (DistanceMeasure::GetMatrix): every other row mixed with
    # its successor; reads one or two topology rows per matrix row, writes N x N floats
    _, _, off = plan.repaint_layout(w)
    mat = torch.empty(N * N, dtype=torch.float32, device="cuda")
    interp = (np.arange(N) & 1).astype(np.uint8)
    half = np.full(N, 0.5)
    ms = []
    for _ in range(reps + 2):
        plan.distance_matrix(N, top.data_ptr(), pitch, ls.data_ptr(), rows, off[:N], interp, half, half, mat.data_ptr(),
                             stream)
        ms.append(plan.timings()["paint_ms"])
    t = min(ms) * 1e-3
    b_dm = 4.0 * N * N + 4.0 * N * N * 1.5
    out["distance_matrix"] = {"ms": min(ms), "roofline": {"bound": "hbm", "achieved": b_dm / t / 1e9, "peak": HBM_PEAK_GBS,
                                                          "unit": "GB/s", "frac": b_dm / t / 1e9 / HBM_PEAK_GBS,
                                                          "traffic": None}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--waves", type=int, default=0, help="waves per target (0 = automatic)")
    ap.add_argument("--variant", type=int, default=1, help="recursion kernel: 1 register-resident, 0 plain HIP cross-check")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive leg (relate_paint_stones on host buffers)")
    ap.add_argument("--repaint", type=int, default=-1, metavar="W",
                    help="also time the re-painting (RePaintSection of all targets) of window W; adds a 'repaint' object")
    ap.add_argument("--theta-data", type=float, default=0.0,
                    help="Data::theta when it differs from the painter's 0.001 (experiments: the Rust caller's two thetas)")
    ap.add_argument("--sharded", action="store_true",
                    help="paint ONE chunk with its targets sharded over the ranks (the default when --gpus > 1)")
    ap.add_argument("--independent-chunks", action="store_true",
                    help="with --gpus > 1: one chunk of the same shape per rank instead (no broadcast)")
    ap.add_argument("--N", type=int, default=0, help="override the number of haplotypes (experiments)")
    ap.add_argument("--L", type=int, default=0, help="override the number of SNPs (experiments)")
    args = ap.parse_args()

    import torch
    import relate_amd as ra

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the painting path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # --sharded on one GPU, not under torch.distributed.run: a group of one rank
            os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 300))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    N, L, desc = WORKLOADS[args.workload]
    if args.N or args.L:
        N, L = args.N or N, args.L or L
        desc = f"custom {N} haplotypes x {L} SNPs"
    theta = 0.001
    theta_data = args.theta_data if args.theta_data > 0 else theta
    sharded = (world > 1 and not args.independent_chunks) or args.sharded
    # One chunk for the whole job (its targets sharded over the ranks), or -- the reference's own way of using
    # several processes, one per --chunk_index -- a chunk of the same shape per rank.
    seq = r = wb = None
    if not sharded or rank == 0:
        seq, r = ra.synth_chunk(N, L, seed=2 + (0 if sharded else rank), skewed_map=(args.workload == "c5"))
        wb = ra.window_boundaries(seq, memory_gb=WORKLOAD_MEMORY_GB.get(args.workload, 5.0))
    if sharded and world > 1:
        box = [wb]
        dist.broadcast_object_list(box, src=0)
        wb = box[0]
    W = len(wb) - 1
    R = ra.record_bytes(N)
    stream = torch.cuda.current_stream().cuda_stream
    kern = {"pack_ms": 0.0, "prepare_ms": 0.0, "paint_ms": 0.0}
    bufs = {}
    plan = None
    shard = None

    if sharded:
        from relate_amd.dist import GpuBackend, ShardedChunk
        be = GpuBackend(local_rank)
        be.plan.close()
        be.plan = ra.Plan(local_rank, args.variant, args.waves)
        plan = be.plan
        d_seq = torch.from_numpy(seq).cuda() if rank == 0 else None
        d_r = torch.from_numpy(r).cuda() if rank == 0 else None
        shard = ShardedChunk(be, N, L, wb, theta, theta_data)
        kern.update({"broadcast_ms": 0.0, "shard_ms": 0.0})

        def step():
            s = shard.step(d_seq, d_r)
            kern["pack_ms"] += shard.last["pack"]
            kern["broadcast_ms"] += shard.last["broadcast"]
            kern["shard_ms"] += shard.last["shard"]
            kern["paint_ms"] += shard.last["paint"]  # prepare + recursion of this rank's range
            return s
    else:
        d_seq = torch.from_numpy(seq).cuda()
        d_r = torch.from_numpy(r).cuda()
        # All stones of the chunk stay in HBM when they fit; otherwise the targets are painted in batches into
        # one buffer (as relate_paint_chunk does before it writes each batch to the files).
        budget = 96 << 30
        kb = max(1, min(N, budget // (W * R)))
        bufs["out"] = torch.empty(W * kb * R, dtype=torch.uint8, device="cuda")  # dropped before the end-to-end leg
        plan = ra.Plan(local_rank, args.variant, args.waves)

        def step():
            plan.load_device(N, L, d_seq.data_ptr(), d_r.data_ptr(), wb, stream)
            s = 0
            for k0 in range(0, N, kb):
                s += plan.prepare(theta, theta_data, k0, min(N, k0 + kb), stream)
                plan.paint(bufs["out"].data_ptr(), stream)
                t = plan.timings()  # HIP events recorded on the launch stream around each stage
                kern["prepare_ms"] += t["prepare_ms"]
                kern["paint_ms"] += t["paint_ms"]
            kern["pack_ms"] += t["pack_ms"]
            return s

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    sum_nd = 0
    for _ in range(args.warmup):
        sum_nd = step()
    fence()
    for k in kern:
        kern[k] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sum_nd = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    for k in kern:
        kern[k] /= max(args.steps, 1)
    ranges = None
    if sharded:
        mine = torch.tensor([shard.range[0], shard.range[1], int(kern["paint_ms"] * 1e3)], dtype=torch.int64, device="cuda")
        if world > 1:
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        ranges = [[int(v[0]), int(v[1]), int(v[2]) / 1e3] for v in allr]

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        chunks = 1 if sharded else world  # chunks painted per step by the whole job
        u_nominal = float(N) * (N - 1) * L
        u_actual = 2.0 * (N - 1) * sum_nd
        # Algorithmic work of the recursion kernel per chunk (all N targets; with the targets sharded, a rank's
        # launch does its share of it).  FP64 flop per state and site: 3 forward (add, multiply, accumulate) + 4
        # backward as the reference writes it (add, fused multiply-add = 2, accumulate): 7, the figure `frac` is
        # priced with.  With one theta for painter and data the kernel runs the backward pass on the emission-
        # weighted state, which takes 3: 6 flop are executed, given beside it as `frac_executed`.
        f_alg = 7.0 * (N - 1) * sum_nd
        f_exec = (6.0 if theta_data == theta else 7.0) * (N - 1) * sum_nd
        b_alg = (2.0 * ((N + 7) // 8) * sum_nd              # one packed haplotype row per site and pass
                 + 32.0 * sum_nd                            # per-gap terms, both passes
                 + float(W) * N * R                         # records written
                 + ((N + 7) // 8) * L + 8.0 * L)            # inputs read once
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and not sharded:
            try:
                tj = json.load(open(tp))
                traffic = tj.get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        geo = plan.geometry()
        if sharded:
            # the slowest rank's recursion launch sets the pace; its share of the chunk's work is what it painted
            t_paint = max(v[2] for v in ranges) * 1e-3
            share = 1.0 / world
            parallelism = (f"one chunk, targets sharded over {world} rank(s) in contiguous ranges of equal numbers of "
                           "derived sites; one broadcast of the packed matrix and one of r per step over RCCL, no reduction")
        else:
            t_paint = kern["paint_ms"] * 1e-3
            share = 1.0
            parallelism = f"{world} independent chunk(s), one per GPU"
        out = {
            "metric": "HMM cell-updates/sec (N^2*L) for Paint",
            "value": u_nominal * chunks / dt * args.steps,
            "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "N": N, "L": L, "windows": W, "theta": theta, **({"theta_data": theta_data} if theta_data != theta else {}),
                       "sum_derived_sites": int(sum_nd), "chunks": chunks,
                       **({"target_ranges": ranges} if sharded else {"target_batches": (N + kb - 1) // kb}),
                       "kernel_geometry": geo, "parallelism": parallelism},
            "actual_state_updates_per_s": u_actual * chunks / dt * args.steps,
            "stage_ms": kern,
            # Compute-bound on the FP64 vector pipe (no MFMA is used: the recurrence is not a contraction), so the
            # bound is named for what binds.  Of the two roofs the benchmark contract lists ("hbm", "mfma") the
            # nearest is the second: vector FP64 peaks at the same 78.6 TFLOP/s as the FP64 matrix pipe.
            "roofline": {"bound": "valu_fp64", "bound_contract_name": "mfma",
                         "pipe": "fp64 vector (v_add/v_mul/v_fma_f64); FMA = 2 flop; no matrix instructions",
                         "kernel": "paint_kernel_r (forward / backward recursion, one workgroup per pass)",
                         "achieved": share * f_alg / t_paint / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": share * f_alg / t_paint / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "frac_executed": share * f_exec / t_paint / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "frac_note": "frac / achieved price the launch at the algorithmic 7 flop per state and site pair "
                                      "(3 forward + 4 backward as the reference writes them); frac_executed is the hardware "
                                      "utilisation: flop the kernel really issued (6 with one theta) over the same time",
                         "flop_per_state_update_pair": 7, "kernel_ms": t_paint * 1e3, "traffic": traffic},
            "roofline_hbm": {"bound": "hbm", "achieved": share * b_alg / t_paint / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": share * b_alg / t_paint / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": share * b_alg,
                             "traffic": traffic},
        }
        if sharded:
            out["roofline"]["note"] = ("per GPU: the slowest rank's prepare + recursion time against its 1/n_gpus share of "
                                       "the chunk's work (the ranges hold equal numbers of derived sites)")
            kb = 0
        if args.repaint >= 0 and not sharded and kb >= N:
            out["repaint"] = repaint_leg(plan, torch, bufs["out"], args.repaint, N, R, stream)
        if world == 1 and not sharded and not args.no_cpu:
            out["cpu_baseline"], out["cpu_baseline_all_cores"], o3 = cpu_baseline(seq, r, wb, theta)
            if o3 is not None:
                out["cpu_baseline_o3"] = o3
        if world == 1 and not sharded and not args.no_e2e:
            bufs.clear()
            torch.cuda.empty_cache()
            out["end_to_end"] = end_to_end(ra, seq, r, wb, theta, theta_data)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
