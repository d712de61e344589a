#!/usr/bin/env python3
"""A chromosome-20-sized painting job end to end on one GPU: a phased data set of 5008 haplotypes and several
hundred thousand SNPs written as .haps / .sample / genetic map, read back through the loader (relate_haps_*: bit rows,
chunks with the 20 000-SNP overlap, windows by the memory rule; src/data.cpp:134-528), and every chunk painted twice:

  * resident: all targets of the chunk in one launch, records left in HBM (what bench.py times): kernel time;
  * end to end: relate_plan_paint_to_files, i.e. batches of targets, records over the link into the pinned ring and
    out through pwrite.  A chunk's window files are ~100 GB, so all but a few sampled windows are written to
    /dev/null; the sampled ones go to real files and sampled targets in them are compared with the CPU oracle.

    python3 tools/chr20_e2e.py [--N 5008] [--snps 660000] [--dir /tmp/chr20] [--keep-windows 3] [--check-targets 3]

Prints one JSON line per chunk and a summary line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def write_data_set(d, N, L, seed, block=40000):
    """.haps (one line per SNP: chromosome, id, position, two alleles, N alleles), .sample (N/2 diploid individuals),
    genetic map (position, rate, cM) with 1 kb spacing at 1 cM/Mb.  Returns the paths and the seconds spent."""
    import relate_amd as ra
    t0 = time.time()
    os.makedirs(d, exist_ok=True)
    haps, sample, gmap = (os.path.join(d, n) for n in ("chr20.haps", "chr20.sample", "chr20.map"))
    with open(sample, "w") as f:
        f.write("ID_1 ID_2 missing\n0 0 0\n")
        for i in range(N // 2):
            f.write(f"ind{i} ind{i} 0\n")
    last_bp = 1000 * (L + 1)
    with open(gmap, "w") as f:
        f.write("pos COMBINED_rate Genetic_Map\n")
        f.write("0 1.0 0.0\n")
        f.write(f"{last_bp} 1.0 {last_bp * 1e-6}\n")
    with open(haps, "wb", buffering=1 << 24) as f:
        done = 0
        while done < L:
            n = min(block, L - done)
            seq, _ = ra.synth_chunk(N, n, seed + done // block)
            txt = np.full((n, 2 * N + 1), ord(" "), dtype=np.uint8)
            txt[:, 1:2 * N:2] = seq
            txt[:, 2 * N] = ord("\n")
            for i in range(n):
                s = done + i
                f.write(b"20 snp%d %d A G" % (s, 1000 * (s + 1)))
                f.write(txt[i].tobytes())
            done += n
    return haps, sample, gmap, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=5008)
    ap.add_argument("--snps", type=int, default=660000)
    ap.add_argument("--dir", default="/tmp/chr20")
    ap.add_argument("--memory", type=float, default=5.0)
    ap.add_argument("--keep-windows", type=int, default=3, help="windows per chunk written to real files and checked")
    ap.add_argument("--check-targets", type=int, default=3, help="targets per kept window compared with the oracle")
    ap.add_argument("--seed", type=int, default=20)
    args = ap.parse_args()
    import torch
    import relate_amd as ra
    from relate_amd.haps import Haps
    import oracle_lib as ol
    from test_gpu_parity import compare_records

    N, theta = args.N, 0.001
    haps, sample, gmap, t_write = write_data_set(args.dir, N, args.snps, args.seed)
    text_gb = os.path.getsize(haps) / 1e9
    t0 = time.time()
    h = Haps(haps, sample, gmap, args.memory)
    t_parse = time.time() - t0
    assert h.N == N and h.L == args.snps
    _, _, _, r_all = h.positions()
    rows_all = h.rows()  # bit rows of the whole data set (for the oracle's byte matrix of sampled chunks)
    R = ra.record_bytes(N)
    print(json.dumps({"data_set": {"N": N, "snps": h.L, "haps_text_gb": round(text_gb, 2), "written_s": round(t_write, 1),
                                   "parsed_s": round(t_parse, 2), "parse_gb_per_s": round(text_gb / t_parse, 2),
                                   "chunks": h.num_chunks}}), flush=True)
    plan = ra.Plan(0)
    st = torch.cuda.current_stream().cuda_stream
    totals = {"kernel_ms": 0.0, "e2e_s": 0.0, "record_gib": 0.0, "cells": 0.0, "sum_nd": 0}
    rng = np.random.default_rng(args.seed)
    devnull = os.open("/dev/null", os.O_WRONLY)
    for c in range(h.num_chunks):
        first, Lc, wb = h.chunk(c)
        W = len(wb) - 1
        t0 = time.time()
        h.load_chunk(plan, c)
        torch.cuda.synchronize()
        t_load = time.time() - t0
        # resident pass: one launch over all targets, timed with the plan's own HIP events
        out = torch.empty(W * N * R, dtype=torch.uint8, device="cuda")
        sum_nd = plan.prepare(theta, theta, 0, N, st)
        plan.paint(out.data_ptr(), st)
        torch.cuda.synchronize()
        tm = plan.timings()
        pack_ms, prep_ms, paint_ms = tm["pack_ms"], tm["prepare_ms"], tm["paint_ms"]
        geo = plan.geometry()
        # end-to-end pass through the library's copy-back: kept windows to files, the rest to /dev/null
        keep = sorted(set([0, W - 1] + [int(v) for v in rng.integers(0, W, max(0, args.keep_windows - 2))]))[:max(1, args.keep_windows)]
        paths = {w: os.path.join(args.dir, f"chunk{c}_relate_{w}.bin") for w in keep}
        fds = [os.open(paths[w], os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644) if w in paths else devnull for w in range(W)]
        t0 = time.time()
        plan.paint_to_files(theta, theta, 0, N, fds)
        t_e2e = time.time() - t0
        for w in keep:
            os.close(fds[w])
        # the kept windows: equal to the resident records, and sampled targets equal to the oracle's
        worst, file_diff = 0.0, 0
        out3 = out.view(W, N, R)
        picks = sorted(set([0, N - 1] + [int(v) for v in rng.integers(0, N, max(0, args.check_targets - 2))]))
        for w in keep:
            got = np.fromfile(paths[w], dtype=np.uint8)
            assert got.size == N * R, "window file has the wrong size"
            file_diff += int(np.count_nonzero(got != out3[w].reshape(-1).cpu().numpy()))
            os.unlink(paths[w])
        if args.check_targets > 0:
            bits = np.unpackbits(rows_all[first:first + Lc].view(np.uint8), axis=1, bitorder="little")[:, :N]
            seq = (bits + ord("0")).astype(np.uint8)
            r = r_all[first:first + Lc].copy()
            for k in picks:
                want = ol.oracle().paint_stones(seq, r, theta, theta, wb, k, k + 1)
                got = out3[:, k].cpu().numpy().reshape(W, 1, R)
                worst = max(worst, compare_records(got, want, N, f"chunk {c} target {k}"))
        del out
        gib = W * N * R / 2**30
        cells = float(N) * (N - 1) * Lc
        line = {"chunk": c, "first_snp": first, "snps": Lc, "windows": W, "sum_derived_sites": int(sum_nd),
                "geometry": geo, "load_s": round(t_load, 3), "pack_ms": round(pack_ms, 2), "prepare_ms": round(prep_ms, 2),
                "kernel_ms": round(paint_ms, 2), "resident_cell_updates_per_s": cells / ((pack_ms + prep_ms + paint_ms) * 1e-3),
                "end_to_end_s": round(t_e2e, 3), "record_gib": round(gib, 1), "end_to_end_gib_per_s": round(gib / t_e2e, 1),
                "end_to_end_cell_updates_per_s": cells / t_e2e, "windows_to_files": keep, "targets_against_oracle": picks if args.check_targets > 0 else [],
                "worst_share_of_floats_not_bit_identical": worst,
                "bytes_of_kept_files_differing_from_resident_records": file_diff}
        print(json.dumps(line), flush=True)
        totals["kernel_ms"] += pack_ms + prep_ms + paint_ms
        totals["e2e_s"] += t_e2e + t_load
        totals["record_gib"] += gib
        totals["cells"] += cells
        totals["sum_nd"] += int(sum_nd)
    os.close(devnull)
    plan.close()
    h.close()
    for p in (haps, sample, gmap):
        os.unlink(p)
    print(json.dumps({"summary": {"chunks": h.num_chunks, "resident_ms": round(totals["kernel_ms"], 1),
                                  "resident_cell_updates_per_s": totals["cells"] / (totals["kernel_ms"] * 1e-3),
                                  "end_to_end_s": round(totals["e2e_s"], 2), "record_gib": round(totals["record_gib"], 1),
                                  "end_to_end_cell_updates_per_s": totals["cells"] / totals["e2e_s"],
                                  "parse_s": round(t_parse, 2), "sum_derived_sites": totals["sum_nd"]}}), flush=True)


if __name__ == "__main__":
    main()
