#!/usr/bin/env python3
"""Build-time check of the register-resident recursion kernels (relate_amd/csrc/paint_regs.hpp).

The kernels keep their state and their row buffers and fixed scalars in registers addressed by number from inline
asm (paint_regs.hpp: v48+ / s52+ unless the kernel states other limits in a "relate_limits" comment; paint_loop.hpp:
v24+ / s24+), and rely on the compiler staying below those limits.  This script reads the device
assembly (hipcc -S --cuda-device-only) of the instantiation files and fails if any instruction the
COMPILER emitted (i.e. outside the asm statements) inside a paint_kernel_r / repaint_kernel_r kernel names v48+ or s52+.

It also fails if the compiler put a scalar load of its own into the latch of a loop (a block inside a loop that falls
through into the loop's header): such a load runs at every site and brings a wait for all scalar loads with it,
the row fetch in flight included (measured: 1.2 % of the 5008-haplotype launch for one reloaded kernel argument).
Loads in the rare paths of a loop body (stones, rescales) sit in conditional blocks and are not meant.

    python tools/check_regs.py relate_amd/csrc/paint_regs_g1.s ...
"""
import re
import sys

# first reserved VGPR / SGPR per kernel family (namespace of the mangled name)
LIMITS = {"regs": (48, 52), "aloop": (24, 24)}
REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]")


def check(path):
    bad = []
    kernel = None
    in_asm = False
    counts = {}
    block_loads, block_in_loop, after_label = [], False, False
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_ZN6relate\d+(regs|aloop)\d+(?:paint|repaint|repaint_fwd|repaint_bwd)_kernel_[ra]\w+):", line)
        if m:
            kernel = m.group(1)
            v_limit, s_limit = LIMITS[m.group(2)]
            a_limit = None  # first reserved accumulation register, where the kernel names one
            counts[kernel] = 0
            continue
        if line.startswith("\t.end_amdhsa_kernel") or ".Lfunc_end" in line:
            kernel = None
        if kernel is None:
            block_loads, block_in_loop, after_label = [], False, False
            continue
        if line.startswith(".LBB"):
            if block_loads and block_in_loop and "Loop Header" in line:
                bad += [f"{path}:{n}: {kernel[:60]}: scalar load at a loop latch: {c}" for n, c in block_loads]
            block_loads, block_in_loop, after_label = [], False, True
            continue
        if after_label:
            block_in_loop = "in Loop: Header=" in line
            after_label = False
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        lim = re.search(r"relate_limits v(\d+) s(\d+)", line)
        if lim and in_asm:  # the kernel states its own limits (claim_registers)
            v_limit, s_limit = int(lim.group(1)), int(lim.group(2))
            continue
        lim = re.search(r"relate_limits a(\d+)", line)
        if lim and in_asm:  # (claim_slots)
            a_limit = int(lim.group(1))
            continue
        code = line.split(";")[0]
        if in_asm or not code.strip() or code.lstrip().startswith("."):
            continue
        counts[kernel] += 1
        if re.match(r"\s+s_load_dword", code):
            block_loads.append((ln, code.strip()))
        for r in REG.finditer(code):
            if r.group(1):
                cls, hi = r.group(1), int(r.group(2))
            else:
                cls, hi = r.group(3), int(r.group(5))
            limit = {"v": v_limit, "s": s_limit, "a": a_limit if a_limit is not None else 1 << 30}[cls]
            if hi >= limit:
                bad.append(f"{path}:{ln}: {kernel[:60]}: {code.strip()}")
    return counts, bad


def main():
    total_bad = []
    nk = 0
    for p in sys.argv[1:]:
        counts, bad = check(p)
        nk += len(counts)
        total_bad += bad
    if nk == 0:
        print("check_regs: no paint_kernel_r kernels found", file=sys.stderr)
        return 2
    if total_bad:
        print("\n".join(total_bad[:40]), file=sys.stderr)
        print(f"check_regs: {len(total_bad)} compiler instructions touch reserved registers or load at a loop latch", file=sys.stderr)
        return 1
    print(f"check_regs: {nk} kernels clean")
    return 0


if __name__ == "__main__":
    sys.exit(main())
