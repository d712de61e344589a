"""Generates the golden fixtures in this directory by running the REFERENCE's own
painting code (oracle/_ref/libpaint_ref.so = /root/reference/src/fast_painting.cpp
compiled by oracle/Makefile) on small seeded inputs.  Run in the container that
holds /root/reference:

    make -C oracle && python tests/golden/make_golden.py

Each .npz holds data only: the inputs (haplotypes bit-packed with
numpy.packbits, r, window boundaries, thetas) and the reference's outputs
(stepping-stone records exactly as they would lie in relate_<w>.bin, and
RePaintSection outputs for a few sections)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402
import synth_np as sy  # noqa: E402

CASES = {
    # name: (make_chunk kwargs, theta_painter, theta_data)
    "n64_flat": (dict(N=64, L=2000, W=4, seed=3, r_mode="flat"), 0.001, 0.001),
    "n70_skew_emptywin": (dict(N=70, L=3000, W=7, seed=4, r_mode="skew", empty_window=True), 0.001, 0.001),
    "n33_mixed_theta": (dict(N=33, L=1200, W=5, seed=2, r_mode="skew"), 0.001, 0.01),
    "n130_tail": (dict(N=130, L=1500, W=6, seed=5, r_mode="flat", empty_window=True), 0.025, 0.025),
    "n16_theta1e-4": (dict(N=16, L=400, W=3, seed=1, r_mode="flat"), 1e-4, 1e-4),
}


def save(name, seq, r, wb, tp, td, extra=None):
    ref = ol.ref()
    out = ref.paint_stones(seq, r, tp, td, wb)
    L, N = seq.shape
    d = dict(N=N, L=L, seq_bits=np.packbits(seq == ord("1")), r=r, wb=wb, theta_painter=tp, theta_data=td,
             stones=out)
    if tp == td:
        # RePaintSection of a few (window, target) pairs, seeded with the stones above
        st = ol.decode_records(out)
        W = len(wb) - 1
        picks = [(0, 0), (W - 1, N - 1), (W // 2, N // 2), (min(1, W - 1), 1)]
        for i, (w, k) in enumerate(picks):
            top, ls = ref.repaint_section(seq, r, td, int(st["a_snp"][w, k]), int(st["b_snp"][w, k]),
                                          st["alpha"][w, k], st["beta"][w, k], float(st["a_ls"][w, k]),
                                          float(st["b_ls"][w, k]), k)
            d[f"rp{i}_wk"] = np.array([w, k], dtype=np.int32)
            d[f"rp{i}_top"] = top
            d[f"rp{i}_ls"] = ls
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, out.shape, f"{out.nbytes/1e3:.0f} kB")


def fast_log_inputs():
    """Arguments for fast_log (src/fast_log.hpp:6-21): the decades of the reference's own test
    (test/test_log.cpp:6-15), every power of ten and of two a float holds, the rescale thresholds the painting
    feeds it (1e-10, 1e10), the neighbours of mantissa and exponent boundaries, zero, denormals, the largest
    finite value, infinity, and seeded log-uniform random values."""
    rng = np.random.default_rng(20)
    parts = [
        np.array([1e-4, 1e-3, 1e-2, 1e-1, 1.0, 1e1, 1e2, 1e3, 1e4], dtype=np.float32),
        (10.0 ** np.arange(-45, 39, dtype=np.float64)).astype(np.float32),
        (2.0 ** np.arange(-149, 128, dtype=np.float64)).astype(np.float32),
        np.array([1e-10, 1e10, 9.99999e-11, 1.00001e10, 0.0, 1.17549435e-38, 1.17549421e-38, 1e-45, 3.4028235e38,
                  np.inf, 0.5, 2.0, 3.0, 1.5, 0.75], dtype=np.float32),
        np.exp(rng.uniform(np.log(1e-38), np.log(3e38), 60000)).astype(np.float32),
        np.exp(rng.uniform(np.log(1e-12), np.log(1e12), 20000)).astype(np.float32),
    ]
    x = np.concatenate(parts)
    edges = np.concatenate([np.nextafter(x[:400], np.float32(0)), np.nextafter(x[:400], np.float32(np.inf))])
    return np.concatenate([x, edges[np.isfinite(edges)]]).astype(np.float32)


def save_fast_log():
    x = fast_log_inputs()
    fl = ol.ref().fast_log
    y = np.array([fl(float(v)) for v in x], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "fast_log.npz"), x_bits=x.view(np.uint32), y_bits=y.view(np.uint32))
    print("fast_log", x.size, "values")


def main():
    if not ol.ref_available():
        raise SystemExit("oracle/_ref/libpaint_ref.so missing: run `make -C oracle` where /root/reference exists")
    save_fast_log()
    for name, (kw, tp, td) in CASES.items():
        seq, r, wb = sy.make_chunk(**kw)
        save(name, seq, r, wb, tp, td)
    # The reference's known-answer toy (test/test_painting.cpp): r = 0 and r = 0.01.
    seq, d = sy.toy_test_painting()
    for tag, rv in (("toy_r0", 0.0), ("toy_r001", 0.01)):
        r = np.full(10, rv)
        wb = np.array([0, 10], dtype=np.int32)
        # RePaintSection over the whole range exactly as the reference test drives it (:98-113)
        N, L, theta = 5, 10, 0.025
        pt = np.float32(theta / (N - 1.0) - (1 - theta) / (N - 1.0))
        pn = np.float32((1 - theta) / (N - 1.0))
        tops, lss = [], []
        for k in range(N):
            ab = np.array([np.float32(float(seq[0, k] > seq[0, n])) * pt + pn for n in range(N)], dtype=np.float32)
            top, ls = ol.ref().repaint_section(seq, r, theta, 0, L - 1, ab, np.ones(N, np.float32), 0.0, 0.0, k)
            tops.append(top)
            lss.append(ls)
        extra = {"expected_d": d}
        for k in range(N):
            extra[f"toy_top{k}"] = tops[k]
            extra[f"toy_ls{k}"] = lss[k]
        save(tag, seq, r, wb, theta, theta, extra)


if __name__ == "__main__":
    main()
