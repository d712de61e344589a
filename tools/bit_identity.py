#!/usr/bin/env python3
"""Fraction of stored floats (stones, re-painted topology) that are not bit-identical to the CPU oracle, per
golden vector and kernel variant (needs a GPU).  The parity bar is a tolerance; this reports how close to
bit-identity the kernels are."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_lib as ol  # noqa: E402
import synth_np as sy  # noqa: E402
from conftest import GOLDEN_NAMES, load_golden  # noqa: E402
from test_gpu_parity import compare_records, gpu_paint  # noqa: E402
from test_gpu_repaint import check_window, gpu_paint_and_repaint  # noqa: E402

for name in GOLDEN_NAMES:
    g = load_golden(name)
    fr = []
    for variant, waves in ((1, 0), (1, 2), (0, 0)):
        got = gpu_paint(g["seq"], g["r"], g["theta_painter"], g["theta_data"], g["wb"], variant, waves)
        fr.append(compare_records(got, g["stones"], g["N"], name))
    print(f"{name:22s} N={g['N']:4d} floats {g['stones'].size // 4:8d}  not identical: v1 {fr[0]:.2e}  v1/2 waves {fr[1]:.2e}  v0 {fr[2]:.2e}")
for (N, L, W, seed, mode, theta) in ((300, 3000, 5, 21, "flat", 0.001), (300, 3000, 5, 22, "skew", 0.001), (1000, 1200, 4, 24, "flat", 0.001)):
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=seed, r_mode=mode, empty_window=True)
    want = ol.oracle().paint_stones(seq, r, theta, theta, wb)
    f1 = compare_records(gpu_paint(seq, r, theta, theta, wb, 1, 0), want, N)
    f0 = compare_records(gpu_paint(seq, r, theta, theta, wb, 0, 0), want, N)
    host, res = gpu_paint_and_repaint(seq, r, theta, wb, [1])
    top, ls, off, pitch = res[1]
    fr = check_window(seq, r, theta, host, 1, top, ls, off, pitch, list(range(0, N, 11)))
    print(f"synthetic N={N} L={L} {mode}: stones not identical v1 {f1:.2e} v0 {f0:.2e}; topology (window 1, sampled targets) {fr:.2e}")
