"""What each rank of an n-GPU run of the benchmark's chunk would do, timed on one GPU: for n in 1, 2, 4, 8 the
targets are cut into n contiguous ranges of equal numbers of derived sites (the cut every rank computes), and each
range is prepared and painted on its own (HIP events around prepare and recursion, as bench.py's sharded path
reports them).  The slowest range of a cut is that run's recursion time per step; the broadcast and the packing
are not in it.  The records of every range are compared with the same targets of the whole-chunk painting.
    python tools/shard_projection.py [workload] > gpurun_out/shard_projection.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import relate_amd as ra  # noqa: E402
from bench import WORKLOADS  # noqa: E402
from relate_amd.dist import shard_by_work  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
N, L, _ = WORKLOADS[name]
seq, r = ra.synth_chunk(N, L, 2, skewed_map=(name == "c5"))
wb = ra.window_boundaries(seq, memory_gb=5.0)
W, R = len(wb) - 1, ra.record_bytes(N)
d_seq, d_r = torch.from_numpy(seq).cuda(), torch.from_numpy(r).cuda()
st = torch.cuda.current_stream().cuda_stream
plan = ra.Plan(0)
plan.load_device(N, L, d_seq.data_ptr(), d_r.data_ptr(), wb, st)
torch.cuda.synchronize()
nd = plan.derived_counts()
full = torch.empty(W * N * R, dtype=torch.uint8, device="cuda")
part = torch.empty(W * ((N + 1) // 2 + 64) * R, dtype=torch.uint8, device="cuda")
base = None
for n in (1, 2, 4, 8):
    bounds = shard_by_work(nd, n)
    worst, lines = 0.0, []
    for g in range(n):
        k0, k1 = int(bounds[g]), int(bounds[g + 1])
        out = full if n == 1 else part
        best = None
        for rep in range(2):
            plan.prepare(0.001, 0.001, k0, k1, st)
            plan.paint(out.data_ptr(), st)
            t = plan.timings()
            ms = t["prepare_ms"] + t["paint_ms"]
            best = ms if best is None else min(best, ms)
        geo = plan.geometry()
        same = ""
        if n > 1:  # the range's records against the same targets of the whole-chunk painting
            a = out[: W * (k1 - k0) * R].view(W, k1 - k0, R)
            b = full.view(W, N, R)[:, k0:k1]
            same = " identical" if bool((a == b).all()) else " DIFFERENT"
        worst = max(worst, best)
        lines.append(f"    range {g}: targets [{k0}, {k1}) {int(nd[k0:k1].sum())} sites, {geo['waves_per_target']} x {geo['states_per_lane']} words, {best:.2f} ms{same}")
    if n == 1:
        base = worst
    print(f"{name} over {n} GPU(s): slowest range {worst:.2f} ms -> {base / worst:.2f}x of one GPU, efficiency {base / worst / n:.2f}")
    print("\n".join(lines), flush=True)
