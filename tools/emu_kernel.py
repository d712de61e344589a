#!/usr/bin/env python3
"""Runs a compiled painting kernel (the assembly the build keeps, relate_amd/csrc/paint_regs_g*.s) on the CPU with
tools/gfx950_emu.py: the chunk is laid out in emulated device memory as the library lays it out in HBM, the
workgroups of chosen targets are interpreted, and the records they write are compared with the reference's.

    python3 tools/emu_kernel.py [--asm FILE] [--kernel REGEX] [--golden NAME] [--waves G] [--mpl M] [--targets a:b]
"""
import argparse
import os
import re
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import emu_paint_loop as lists  # noqa: E402  (the chunk's per-target lists, restated)
import gfx950_emu as emu  # noqa: E402


def build_memory(ch, targets):
    """Device buffers of a prepared launch over `targets` (a list of haplotype indices, painted as K targets with
    k_begin = targets[0]; they must be consecutive) and the kernel-argument block.  Returns (mem, kernarg, out_name)."""
    k0, K = targets[0], len(targets)
    assert targets == list(range(k0, k0 + K))
    N, L, W = ch.N, ch.L, ch.W
    mem = emu.Memory()

    def alloc(name, data, writable=False):
        n = len(data)
        return mem.add(name, data, n + n // 8 + 256, writable)  # DevBuf::reserve's headroom

    rows_alloc = (L * ch.stride + 8 * 80 + 64) * 8
    rows = mem.add("rows", ch.rows.tobytes(), rows_alloc)
    dk_b, rec_b, off_l, nd_l = bytearray(), bytearray(), [], []
    o = 0
    for k in targets:
        d = ch.dk[k]
        off_l.append(o)
        nd_l.append(len(d))
        o += len(d)
        dk_b += np.asarray(d, dtype=np.int32).tobytes()
        for (cf, nx, f2, b2) in ch.rec[k]:
            rec_b += struct.pack("<ddii", cf, nx, f2, b2)
    dk = alloc("dk", dk_b)
    rec = alloc("rec", rec_b)
    off = alloc("off", np.asarray(off_l, dtype=np.int64).tobytes())
    nd = alloc("nd", np.asarray(nd_l, dtype=np.int32).tobytes())
    bbj = alloc("bbj", np.ascontiguousarray(ch.bbj[k0:k0 + K], dtype=np.int32).tobytes())
    bej = alloc("bej", np.ascontiguousarray(ch.bej[k0:k0 + K], dtype=np.int32).tobytes())
    order = alloc("order", np.argsort(-np.asarray(nd_l), kind="stable").astype(np.int32).tobytes())
    wb = alloc("wb", np.asarray(ch.wb, dtype=np.int32).tobytes())
    R = 56 + 8 * N
    out = mem.add("out", bytes(W * K * R), W * K * R, writable=True)
    c = ch.c
    consts = struct.pack("<13dii", c["Nm1"], c["prior_mis"], c["prior_match"], c["rho"], c["log_ntheta"], c["log_small"],
                         c["log_Nm1"], c["theta_d"], c["ntheta_d"], c["inv_theta_d"], c["inv_ntheta_d"], c["lo"], c["hi"], c["same_theta"], 0)
    ka = struct.pack("<11Q6i", rows, dk, rec, off, nd, bbj, bej, order, wb, out, 0, ch.stride, N, L, W, k0, K) + consts
    kernarg = mem.add("kernarg", ka, len(ka))
    return mem, kernarg


UNDEFINED = {}  # (instruction index, text) -> registers read before anything wrote them


def run_paint(asm, kernel_rx, ch, targets, G, MPL, verbose=True, garbage=None):
    kern = emu.load_kernel(asm, kernel_rx)
    mem, kernarg = build_memory(ch, targets)
    lds_bytes = G * MPL * 256 + 3 * G * 8
    stats = {}
    for block in range(2 * len(targets)):
        waves = emu.run_workgroup(kern, mem, kernarg, block, G, lds_bytes, garbage=None if garbage is None else garbage + block)
        for w in waves:
            for k, v in w.stats.items():
                stats[k] = stats.get(k, 0) + v
            for key, regs in w.undefined_reads.items():
                UNDEFINED.setdefault(key, set()).update(regs)
        if verbose:
            print(f"  workgroup {block} done", flush=True)
    R = 56 + 8 * ch.N
    out = np.frombuffer(bytes(mem.buffer("out")), dtype=np.uint8).reshape(ch.W, len(targets), R)
    return out, stats, kern


def paint_compiled(seq, r, wb, theta_p, theta_d, targets, G, MPL, asm=None, kernel_rx=None):
    """Records (W, K, 56 + 8N) of `targets` as the compiled kernel of that geometry writes them."""
    asm = asm or os.path.join(ROOT, "relate_amd", "csrc", f"paint_regs_g{G}.s")
    ch = lists.Chunk(seq, r, wb, theta_p, theta_d)
    if G * MPL * 64 < ch.N or G * MPL > ch.stride:
        raise ValueError("geometry does not fit the panel")
    out, stats, kern = run_paint(asm, kernel_rx or rf"paint_kernel_rILi{MPL}ELi{G}E", ch, list(targets), G, MPL, verbose=False)
    return out, stats


def repaint_block_rows(MPL):
    """Rows between two alpha checkpoints (relate_amd/csrc/paint_kernels.h: repaint_block_rows)."""
    slots = (256 - 32 - (56 + 2 * MPL)) // (2 * MPL)
    return 1 if slots < 2 else min(slots, 8)


def repaint_compiled(seq, r, wb, theta, window, stones, targets, G, MPL, asm=None):
    """The two compiled re-painting kernels (repaint_kernel_r<MPL, G, false / true>) interpreted for `targets` of one
    window.  stones: the (K, 56 + 8N) records of that window for those targets (as the painting wrote them).
    Returns (topology (rows, pitch) float32, logscales (rows,) float32, row offsets (K + 1))."""
    import math
    asm = asm or os.path.join(ROOT, "relate_amd", "csrc", f"paint_regs_g{G}.s")
    ch = lists.Chunk(seq, r, wb, theta, theta)
    targets = list(targets)
    k0, K = targets[0], len(targets)
    N, L, W = ch.N, ch.L, ch.W
    if G * MPL * 64 < N or G * MPL > ch.stride:
        raise ValueError("geometry does not fit the panel")
    mem, _ = build_memory(ch, targets)
    pitch = G * MPL * 64
    B = repaint_block_rows(MPL)
    off = np.zeros(K + 1, dtype=np.int64)
    coef, nx = np.zeros(K), np.zeros(K)
    c = ch.c
    for t, k in enumerate(targets):
        off[t + 1] = off[t] + int(ch.bej[k, window] - ch.bbj[k, window] + 1)
        rsum = float(r[ch.dk[k][int(ch.bej[k, window])]])
        nxv = -rsum + c["log_ntheta"]
        rp = 1.0 - math.exp(-rsum)
        if rp > 0.99:
            rp, nxv = 0.99, c["log_small"] + c["log_ntheta"]
        coef[t], nx[t] = rp / ((1.0 - rp) * c["Nm1"]), nxv
    rows = int(off[K])
    cp_rows = rows // B + K + 1
    R = 56 + 8 * N
    stones = np.ascontiguousarray(stones, dtype=np.uint8).reshape(K, R)
    a = {name: mem.add(name, data, len(data), wr) for name, data, wr in (
        ("stones", stones.tobytes(), False), ("row_off", off[:K].tobytes(), False), ("last_coef", coef.tobytes(), False),
        ("last_nx", nx.tobytes(), False), ("scratch", bytes(8 * cp_rows * pitch), True), ("cp_sum", bytes(8 * cp_rows), True),
        ("topology", bytes(4 * rows * pitch), True), ("logscales", bytes(4 * rows), True))}
    g = {n: mem.address(n) for n in ("rows", "dk", "rec", "off", "nd", "bbj", "bej", "order")}
    consts = struct.pack("<13dii", c["Nm1"], c["prior_mis"], c["prior_match"], c["rho"], c["log_ntheta"], c["log_small"],
                         c["log_Nm1"], c["theta_d"], c["ntheta_d"], c["inv_theta_d"], c["inv_ntheta_d"], c["lo"], c["hi"], c["same_theta"], 0)
    ka = struct.pack("<16Q2i6i", g["rows"], g["dk"], g["rec"], g["off"], g["nd"], g["bbj"], g["bej"], g["order"], a["stones"],
                     a["row_off"], a["last_coef"], a["last_nx"], a["scratch"], a["cp_sum"], a["topology"], a["logscales"],
                     window, pitch, ch.stride, N, L, W, k0, K) + consts
    kernarg = mem.add("repaint_kernarg", ka, len(ka))
    stats = {}
    for bwd in (0, 1):
        kern = emu.load_kernel(asm, rf"repaint_{'bwd' if bwd else 'fwd'}_kernel_rILi{MPL}ELi{G}E")
        for block in range(K):
            # LDS as the launcher sizes it: the block sums, and behind them the row the backward launch parks (blocks of > 1 row)
            lds = 2 * G * 8 + (G * MPL * 512 if bwd and repaint_block_rows(MPL) > 1 else 0)
            for w in emu.run_workgroup(kern, mem, kernarg, block, G, lds):
                for key, v in w.stats.items():
                    stats[key] = stats.get(key, 0) + v
    top = np.frombuffer(bytes(mem.buffer("topology")), dtype=np.float32).reshape(rows, pitch)
    ls = np.frombuffer(bytes(mem.buffer("logscales")), dtype=np.float32)
    return top, ls, off, stats


def main():
    from conftest import load_golden
    import oracle_lib as ol
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=os.path.join(ROOT, "relate_amd", "csrc", "paint_regs_g1.s"))
    ap.add_argument("--kernel", default="")
    ap.add_argument("--golden", default="n130_tail")
    ap.add_argument("--waves", type=int, default=1)
    ap.add_argument("--mpl", type=int, default=4)
    ap.add_argument("--targets", default="0:2")
    ap.add_argument("--garbage", type=int, default=None, help="seed: registers, LDS and scratch start with random contents")
    args = ap.parse_args()
    g = load_golden(args.golden)
    a, b = (int(v) for v in args.targets.split(":"))
    targets = list(range(a, b))
    ch = lists.Chunk(g["seq"], g["r"], g["wb"], g["theta_painter"], g["theta_data"])
    rx = args.kernel or rf"paint_kernel_rILi{args.mpl}ELi{args.waves}E"
    try:
        out, stats, kern = run_paint(args.asm, rx, ch, targets, args.waves, args.mpl, garbage=args.garbage)
    except emu.EmuError as e:
        print("FAULT:", e)
        sys.exit(1)
    want = g["stones"][:, a:b]
    d, w = ol.decode_records(out), ol.decode_records(want)
    bad = 0
    for key in d:
        if d[key].dtype.kind == "f":
            nb = int(np.count_nonzero(d[key].view(np.uint32) != w[key].view(np.uint32)))
            if nb:
                print(f"  {key}: {nb} of {d[key].size} floats not bit-identical, max rel {np.max(np.abs(d[key] - w[key]) / np.maximum(np.abs(w[key]), 1e-300)):.2e}")
            bad += nb
        else:
            assert np.array_equal(d[key], w[key]), key
    for (pc, line), regs in sorted(UNDEFINED.items()):
        print(f"  reads a register nothing has written: [{pc}] {line}   <- {', '.join(sorted(regs))}")
    n_instr = sum(stats.values())
    print(f"{kern.name}: {len(targets)} targets, {n_instr} instructions interpreted, every access inside its buffer; "
          f"discrete fields identical, {bad} floats not bit-identical to the reference")


if __name__ == "__main__":
    main()
