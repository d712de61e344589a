#!/usr/bin/env python3
"""Prints the clobber-list macros of relate_amd/csrc/paint_regs.hpp for a register map.

    python tools/gen_clobbers.py <first reserved VGPR> <first state VGPR> <first reserved SGPR>

The lists must be string literals (inline-asm clobbers), hence generated text: RELATE_VCLOB_<MPL> names
v[first reserved] .. v[first state + 2*MPL - 1] cumulatively, RELATE_SCLOB names s[first reserved] .. s101."""
import sys

MPLS = [4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80]


def main():
    cv, a0, cs = (int(x) for x in sys.argv[1:4])
    prev, lo = None, cv
    for m in MPLS:
        hi = a0 + 2 * m
        body = ", ".join(f'"v{i}"' for i in range(lo, hi))
        head = f"RELATE_VCLOB_{prev}, " if prev else ""
        print(f"#define RELATE_VCLOB_{m} {head}{body}")
        prev, lo = m, hi
    print("#define RELATE_SCLOB " + ", ".join(f'"s{i}"' for i in range(cs, 102)))


if __name__ == "__main__":
    main()
