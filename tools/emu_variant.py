#!/usr/bin/env python3
"""Interprets the kernels of a variant build (tools/build_variant.sh NAME ...) on the CPU against the oracle:
    python3 tools/emu_variant.py NAME G [MPL N L k0,k0,... ]...
Default cases per G cover small and c3-sized panels, targets at both ends and across a word."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import emu_kernel as ek  # noqa: E402
import oracle_lib as ol  # noqa: E402
import synth_np as sy  # noqa: E402

CASES = {
    1: [(600, 12, 120, (0, 63, 598)), (2000, 32, 150, (0, 1998))],
    2: [(600, 8, 120, (0, 63, 598)), (5008, 40, 150, (0, 2500, 5006))],
    4: [(600, 4, 120, (0, 63, 598)), (4900, 20, 300, (0, 4898)), (5008, 20, 200, (2500, 5006))],
    8: [(4000, 8, 120, (0, 63, 3998)), (10000, 20, 60, (0, 9998))],
}


def bad_floats(got, want):
    g, w = ol.decode_records(got), ol.decode_records(want)
    bad = 0
    for key in g:
        a, b = g[key], w[key]
        if a.dtype.kind == "f":
            bad += int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32)))
        else:
            assert np.array_equal(a, b), key
    return bad


def main():
    name, G = sys.argv[1], int(sys.argv[2])
    thetas = [(0.001, 0.001)]
    if "--two-theta" in sys.argv:
        thetas.append((0.001, 0.01))
    asm = os.path.join(ROOT, "relate_amd", "csrc", "exp", name, f"paint_regs_g{G}.s")
    if name == "regular":
        asm = os.path.join(ROOT, "relate_amd", "csrc", f"paint_regs_g{G}.s")
    for (N, MPL, L, ks) in CASES[G]:
        seq, r, wb = sy.make_chunk(N=N, L=L, W=2, seed=N + G, r_mode="skew")
        for tp, td in thetas:
            for k0 in ks:
                t0 = time.time()
                want = ol.oracle().paint_stones(seq, r, tp, td, wb, k0, k0 + 2)
                out, stats = ek.paint_compiled(seq, r, wb, tp, td, range(k0, k0 + 2), G, MPL, asm=asm)
                bad = bad_floats(out, want)
                print(f"{name} G={G} N={N} MPL={MPL} k0={k0} thetas={tp},{td}: {bad} floats differ, "
                      f"{sum(stats.values())} instructions, {time.time() - t0:.1f}s", flush=True)
                assert bad == 0
    print(name, G, "OK")


if __name__ == "__main__":
    main()
