import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_lib as ol, synth_np as sy
from test_gpu_parity import gpu_paint
np.set_printoptions(linewidth=200, precision=6)
N, L = int(sys.argv[1]), int(sys.argv[2])
seq, r, _ = sy.make_chunk(N=N, L=L, W=1, seed=5, r_mode=(sys.argv[3] if len(sys.argv) > 3 else "flat"))
wb = np.arange(L + 1, dtype=np.int32)
want = ol.decode_records(ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb))
for variant in (1,):
    d = ol.decode_records(gpu_paint(seq, r, 0.001, 0.001, wb, variant, 0))
    k = int(sys.argv[4]) if len(sys.argv) > 4 else int(np.argmax((seq == 49).sum(0)))
    print("== variant", variant, "target", k, "seq col", "".join(chr(c) for c in seq[:, k]))
    for w in range(L):
        ga, wa = d["alpha"][w, k].astype(np.float64), want["alpha"][w, k].astype(np.float64)
        gb, wbt = d["beta"][w, k].astype(np.float64), want["beta"][w, k].astype(np.float64)
        ea = np.max(np.abs(ga - wa) / np.maximum(wa, 1e-300)); eb = np.max(np.abs(gb - wbt) / np.maximum(wbt, 1e-300))
        print(f"w{w:2d} a_snp {d['a_snp'][w,k]:3d} a_ls {d['a_ls'][w,k]:10.4f}/{want['a_ls'][w,k]:10.4f} relerr_a {ea:9.2e} | b_snp {d['b_snp'][w,k]:3d} b_ls {d['b_ls'][w,k]:10.4f}/{want['b_ls'][w,k]:10.4f} relerr_b {eb:9.2e}")
        if variant == 1 and ea > 1e-6 and w < 6:
            print("    got ", ga[:8]); print("    want", wa[:8])
