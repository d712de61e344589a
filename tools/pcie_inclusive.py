import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import relate_amd as ra
N, L = 2000, 50000
seq, r = ra.synth_chunk(N, L, 2)
wb = ra.window_boundaries(seq, 5.0)
for rep in range(3):
    t0 = time.perf_counter()
    out = ra.paint_stones(seq, r, 0.001, 0.001, wb)
    dt = time.perf_counter() - t0
    print(f"relate_paint_stones host in/out: {dt*1e3:.1f} ms  -> {N*(N-1)*L/dt:.3e} cell-updates/s, {out.nbytes/1e6:.0f} MB of records")
import tempfile, os
d = tempfile.mkdtemp(dir="gpurun_out")
ra.write_chunk(d, 0, seq, r, wb)
for rep in range(2):
    t0 = time.perf_counter()
    ra.paint_chunk(d, 0, 0.001, 1.0, device=0)
    dt = time.perf_counter() - t0
    print(f"relate_paint_chunk files in/out: {dt*1e3:.1f} ms")
import shutil; shutil.rmtree(d)
