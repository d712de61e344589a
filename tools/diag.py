import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_lib as ol
from conftest import load_golden
from test_gpu_parity import gpu_paint
name = sys.argv[1] if len(sys.argv) > 1 else "n16_theta1e-4"
g = load_golden(name)
w = ol.decode_records(g["stones"])
for variant, waves in [(0, 0), (1, 0), (1, 2)]:
    try:
        got = gpu_paint(g["seq"], g["r"], g["theta_painter"], g["theta_data"], g["wb"], variant, waves)
    except Exception as e:
        print("variant", variant, waves, "ERR", e); continue
    d = ol.decode_records(got)
    print("== variant", variant, "waves", waves)
    for key in ("alpha", "beta"):
        gv, wv = d[key].astype(np.float64), w[key].astype(np.float64)
        rel = np.abs(gv - wv) / np.maximum(np.abs(wv), 1e-300)
        rel[wv == 0] = np.abs(gv[wv == 0])
        print(key, "max rel per window", np.round(rel.max(axis=(1, 2)), 12))
        W = gv.shape[0]
        for wi in range(min(W, 3)):
            k = 1
            print("  w", wi, "k", k, "got", gv[wi, k, :6], "want", wv[wi, k, :6])
    print("a_ls got", d["a_ls"][:, 1], "want", w["a_ls"][:, 1])
    print("b_ls got", d["b_ls"][:, 1], "want", w["b_ls"][:, 1])
