#!/usr/bin/env python3
"""Times launches of K < N targets (paint stage only) for every wave count per target.

Usage: geometry_sweep_k.py OUT N1,N2,... [W1,W2,...] ; one line per (N, K, waves asked): geometry used, ms.
Wave counts default to 0,1,2,4,8 (0 = the automatic choice)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import relate_amd as ra

out = sys.argv[1]
sizes = [int(x) for x in sys.argv[2].split(",")]
wave_counts = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 4, 8]
L = 50000
with open(out, "a") as f:
    for N in sizes:
        seq, r = ra.synth_chunk(N, L, seed=2)
        wb = ra.window_boundaries(seq, memory_gb=5.0)
        d_seq, d_r = torch.from_numpy(seq).cuda(), torch.from_numpy(r).cuda()
        d_out = torch.empty((len(wb) - 1) * N * ra.record_bytes(N), dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for K in (N, N // 2, N // 4, N // 8, N // 16):
            for waves in wave_counts:
                plan = ra.Plan(0, 1, waves)
                plan.load_device(N, L, d_seq.data_ptr(), d_r.data_ptr(), wb, stream)
                ms = []
                for _ in range(3):
                    plan.prepare(0.001, 0.001, 0, K, stream)
                    plan.paint(d_out.data_ptr(), stream)
                    ms.append(plan.timings()["paint_ms"])
                g = plan.geometry()
                plan.close()
                line = f"{N} {K} {waves} {g['waves_per_target']}x{g['states_per_lane']} {min(ms):.3f}"
                f.write(line + "\n")
                f.flush()
                print(line, flush=True)
