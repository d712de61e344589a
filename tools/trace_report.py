#!/usr/bin/env python3
"""Summary of a RELATE_TRACE file (4 x u64 per workgroup of the register-resident kernel):
where the workgroups ran and how long each SIMD was busy."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
hw = (a[:, 0] & 0xFFFFFFFF).astype(np.int64)
xcc = (a[:, 0] >> 32).astype(np.int64) & 0xF
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
t0 = a[:, 1].astype(np.int64)
t1 = a[:, 2].astype(np.int64)
ps = (a[:, 3] & 0xFF).astype(np.int64)
nd = (a[:, 3] >> 8).astype(np.int64)
base = t0.min()
dur = (t1 - t0) / 100.0  # us (100 MHz)
print(f"workgroups {len(a)}  span {(t1.max() - base) / 100.0:.1f} us  start spread {(t0.max() - base) / 100.0:.1f} us")
for p in (0, 1):
    m = ps == p
    print(f"pass {p}: n={m.sum()} duration us mean {dur[m].mean():.1f} min {dur[m].min():.1f} max {dur[m].max():.1f}; "
          f"ns per site-step {1e3 * dur[m].sum() / nd[m].sum():.1f}")
key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
u, inv = np.unique(key, return_inverse=True)
cnt = np.bincount(inv)
nb = np.bincount(inv, weights=(ps == 1))
end = np.zeros(len(u))
np.maximum.at(end, inv, (t1 - base) / 100.0)
print(f"SIMDs used {len(u)}; waves per SIMD min {cnt.min()} max {cnt.max()}; histogram {np.bincount(cnt)}")
print("backward waves per SIMD histogram", np.bincount(nb.astype(int)))
for k in range(int(nb.max()) + 1):
    m = nb == k
    if m.any():
        print(f"  SIMDs with {k} backward waves: {m.sum():4d}  mean end {end[m].mean():.0f} us  max end {end[m].max():.0f} us  mean waves {cnt[m].mean():.2f}")
print("per XCC: waves", np.bincount(xcc), " backward", np.bincount(xcc, weights=(ps == 1)).astype(int))
print("first 24 blocks -> (xcc,se,sh,cu,simd,pass):", [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i]), int(ps[i])) for i in range(24)])
