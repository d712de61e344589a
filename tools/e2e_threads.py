"""End-to-end time of relate_paint_stones (host buffers in, records back into pageable host memory) on the
benchmark's chunk for the number of copy-back threads given by RELATE_COPY_THREADS:
    RELATE_COPY_THREADS=8 python tools/e2e_threads.py [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (one HIP runtime per process)
import relate_amd as ra  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
N, L, _ = WORKLOADS[name]
seq, r = ra.synth_chunk(N, L, 2, skewed_map=(name == "c5"))
wb = ra.window_boundaries(seq, memory_gb=5.0)
ra.paint_stones(seq, r, 0.001, 0.001, wb, 0, min(N, 64))
best = None
for rep in range(2):
    t0 = time.perf_counter()
    out = ra.paint_stones(seq, r, 0.001, 0.001, wb)
    dt = time.perf_counter() - t0
    gib = out.nbytes / 2**30
    del out
    best = dt if best is None else min(best, dt)
    print(f"threads={os.environ.get('RELATE_COPY_THREADS', 'auto')} {name}: {dt * 1e3:.0f} ms for {gib:.1f} GiB = {gib / dt:.1f} GiB/s", flush=True)
