#!/usr/bin/env python3
"""Where the floats of a painted target differ from the CPU oracle's: per window and block, the number of floats
whose last bit differs and the relative difference of the doubles behind them (as far as floats show it).
    python3 tools/diag_bit_flips.py N L seed memory_gb k [k ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
import relate_amd as ra  # noqa: E402


def main():
    import torch
    N, L, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    mem = float(sys.argv[4])
    ks = [int(v) for v in sys.argv[5:]]
    seq, r = ra.synth_chunk(N, L, seed)
    wb = ra.window_boundaries(seq, memory_gb=mem)
    W, R = len(wb) - 1, ra.record_bytes(N)
    d_seq = torch.from_numpy(seq).cuda()
    d_r = torch.from_numpy(r).cuda()
    plan = ra.Plan(0)
    st = torch.cuda.current_stream().cuda_stream
    plan.load_device(N, L, d_seq.data_ptr(), d_r.data_ptr(), wb, st)
    for k in ks:
        out = torch.empty(W * R, dtype=torch.uint8, device="cuda")
        plan.prepare(0.001, 0.001, k, k + 1, st)
        plan.paint(out.data_ptr(), st)
        torch.cuda.synchronize()
        got = ol.decode_records(out.cpu().numpy().reshape(W, 1, R))
        want = ol.decode_records(ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb, k, k + 1))
        print(f"target {k}: geometry {plan.geometry()}")
        for name, ls in (("alpha", "a_ls"), ("beta", "b_ls")):
            g, w = got[name][:, 0], want[name][:, 0]
            flips = (g.view(np.uint32) != w.view(np.uint32)).sum(axis=1)
            lsf = got[ls][:, 0].view(np.uint32) != want[ls][:, 0].view(np.uint32)
            print(f"  {name}: {int(flips.sum())} floats differ in {int((flips > 0).sum())} of {W} windows; logscales differ in {int(lsf.sum())}")
            idx = np.nonzero(flips)[0]
            for wi in idx[:40]:
                nz = w[wi] != 0
                rel = (g[wi][nz].astype(np.float64) - w[wi][nz]) / w[wi][nz]
                print(f"    window {wi}: {int(flips[wi])} flips; mean signed rel diff {rel.mean():+.3e}, max |rel| {np.abs(rel).max():.3e}")
                snp = int(want["a_snp" if name == "alpha" else "b_snp"][wi, 0])
                bad = np.nonzero(g[wi].view(np.uint32) != w[wi].view(np.uint32))[0]
                der = seq[snp] == ord("1")
                print(f"      stone at site {snp}; target derived there: {bool(der[k])}; derived count {int(der.sum())}")
                print("      n (word, lane) derived? got want: " + "; ".join(
                    f"{n} ({n // 64},{n % 64}) {int(der[n])} {g[wi][n]:.9e} {w[wi][n]:.9e}" for n in bad[:24]))
                vals = w[wi][bad]
                print(f"      flipped values span {vals.min():.3e} .. {vals.max():.3e}; all values span {w[wi][nz].min():.3e} .. {w[wi].max():.3e}")
                print(f"      flipped among derived: {int(der[bad].sum())} of {len(bad)}; signs of the differences: +{int((g[wi][bad] > w[wi][bad]).sum())} -{int((g[wi][bad] < w[wi][bad]).sum())}")


if __name__ == "__main__":
    main()
