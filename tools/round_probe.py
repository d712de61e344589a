#!/usr/bin/env python3
"""Finds how many targets fit one round of waves: times launches of K targets around the expected limit.

Usage: round_probe.py OUT N WAVES K1,K2,...   (one line per K: geometry, paint ms)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import relate_amd as ra

out, N, waves = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
Ks = [int(x) for x in sys.argv[4].split(",")]
L = 50000
seq, r = ra.synth_chunk(N, L, seed=2)
wb = ra.window_boundaries(seq, memory_gb=5.0)
d_seq, d_r = torch.from_numpy(seq).cuda(), torch.from_numpy(r).cuda()
d_out = torch.empty((len(wb) - 1) * N * ra.record_bytes(N), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
with open(out, "a") as f:
    for K in Ks:
        plan = ra.Plan(0, 1, waves)
        plan.load_device(N, L, d_seq.data_ptr(), d_r.data_ptr(), wb, stream)
        ms = []
        for _ in range(3):
            plan.prepare(0.001, 0.001, 0, K, stream)
            plan.paint(d_out.data_ptr(), stream)
            ms.append(plan.timings()["paint_ms"])
        g = plan.geometry()
        plan.close()
        line = f"{N} {K} {g['waves_per_target']}x{g['states_per_lane']} {min(ms):.3f}"
        f.write(line + "\n")
        f.flush()
