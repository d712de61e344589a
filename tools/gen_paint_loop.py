#!/usr/bin/env python3
"""Generates relate_amd/csrc/paint_loop_gen.hpp: the steady-state loop of the painting recursion as ONE
gfx950 assembly statement per kernel geometry (MPL words per lane, G waves per target).

    python3 tools/gen_paint_loop.py > relate_amd/csrc/paint_loop_gen.hpp      (the Makefile does this)

tools/emu_paint_loop.py imports this module and interprets loop_body() on the CPU with every address checked.

Why generated text: every value the loop carries from site to site lives in a register that is named by number
(states, sums, pointers, the prefetched row words and gap terms), the sweep over the MPL words is unrolled, and the
order of the instructions is the schedule (where the scalar fetches are issued relative to the single wait that
covers them).  None of it is left to the compiler, which sees one opaque statement and keeps to registers
below v[CV] / s[CS].

One site of the one-row recursion  state' = (state + rx) * (mismatch ? rho : 1),  S = sum(state'):

    rx = S * coef                       ls += nx                    lanes without a state <- -rx
    per round of at most B words (B = words the row-word registers hold):
        additions of the round          (no masks needed)
        s_waitcnt lgkmcnt(0)            (row words of the round; in the first round also the next gap terms)
        masked multiplications          (EXEC = mismatching lanes, straight from the row words)
        fetch: next round of this row, or first round of the next row + gap terms of the site after + a vector
               touch of the row two sites ahead (keeps it in L2)
        accumulation of the round       (two accumulators)
    wave reduction (DPP), G > 1: exchange through LDS with one barrier
    leave the loop when the sum is not surely inside the rescale thresholds, or a stone is due at this site
    (the caller handles both in C++ and re-enters), or the list is finished.

The loop runs in either direction over the list (forward pass, or the backward pass on the emission-weighted
state): the direction is data (s[DIR], s[RECSTEP])."""
import sys

MPLS = [4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80]
GS = [1, 2, 4, 8]

CV = 24   # compiler: v0 .. v[CV-1]
B = 16    # row words the SGPR buffer holds
CS = 24   # compiler: s0 .. s[CS-1]

# ---- VGPR map
V = {}
for i, name in enumerate(["RX", "S", "X", "T", "LS", "RHO"]):
    V[name] = CV + 2 * i
V["JUNK"] = CV + 12
V["LANEB"] = CV + 13
V["LDSW"] = CV + 14
V["LDSR"] = CV + 15
A0 = CV + 16

# ---- SGPR map
S = {}
_n = CS


def _s(name, count, align=1):
    global _n
    _n = (_n + align - 1) // align * align
    S[name] = _n
    _n += count


_s("ROWSW", 2, 2)     # first row word of this wave, row 0
_s("STRIDE", 1)       # bytes per packed row
_s("DIR", 1)          # 0: the list is walked upwards (forward pass), 1: downwards
_s("REC", 2, 2)       # running pointer into the site records
_s("RECSTEP", 2, 2)   # +24 / -24 as a 64-bit value
_s("RECBUF", 4, 4)    # landing zone of a record: coef, nx
_s("DKF", 1)          # ... dk two steps up
_s("DKB", 1)          # ... dk two steps down
_s("COEF", 2, 2)      # gap terms of the step to the next site
_s("NX", 2, 2)
_s("DK2", 1)
_s("LEFT", 1)         # sites still to come after this one
_s("STONELEFT", 1)    # value of LEFT at which the next stone is due
_s("SHI", 1)          # high word of the sum
_s("SKEDGE", 2, 2)    # the target's own allele at the forced end site, as a mask
_s("SKM", 2, 2)       # ... of the current site
_s("SELFM", 2, 2)     # lanes without a state: see paint_loop.hpp
_s("PHM", 2, 2)
_s("SELFIDX", 1)
_s("PHIDX", 1)
_s("CNT", 1)
_s("HASSL", 1)
_s("RP", 2, 2)        # first word of this wave in the current row, the next row
_s("RPNEXT", 2, 2)
_s("TMP", 2, 2)       # ... and the row after that
_s("RANK", 1)
_s("T2", 1)
_s("BUF", 2 * B, 4)
assert _n <= 102, _n


def vp(name, off=0):
    return f"v[{V[name] + off}:{V[name] + off + 1}]"


def v1(name, off=0):
    return f"v{V[name] + off}"


def sp(name, off=0):
    return f"s[{S[name] + off}:{S[name] + off + 1}]"


def s1(name, off=0):
    return f"s{S[name] + off}"


def a(i):
    return f"v[{A0 + 2 * i}:{A0 + 2 * i + 1}]"


def rounds(mpl):
    nr = (mpl + B - 1) // B
    sizes = [mpl // nr + (1 if r < mpl % nr else 0) for r in range(nr)]
    firsts = [sum(sizes[:r]) for r in range(nr)]
    return list(zip(firsts, sizes))


def sload(ptr, first, nw):
    """words [first, first + nw) of the row at s[ptr] -> s[BUF ..]; loads of 8 or 4 words (over-reading is
    covered by the slack after the last packed row)"""
    out = []
    done = 0
    while done < nw:
        rest = nw - done
        if rest > 4:
            out.append(f"s_load_dwordx16 s[{S['BUF'] + 2 * done}:{S['BUF'] + 2 * done + 15}], {sp(ptr)}, {8 * (first + done)}")
            done += 8
        else:
            out.append(f"s_load_dwordx8 s[{S['BUF'] + 2 * done}:{S['BUF'] + 2 * done + 7}], {sp(ptr)}, {8 * (first + done)}")
            done += 4
    assert 2 * done <= 2 * B + 0 or True
    return out


def fetch_record():
    """The record at s[REC] into its landing registers (coef, nx, dk two steps up, dk two steps down), and the
    pointer on to the next record of the pass: what the loop expects to be on its way when it is entered."""
    return [f"s_load_dwordx4 s[{S['RECBUF']}:{S['RECBUF'] + 3}], {sp('REC')}, 0",
            f"s_load_dwordx2 s[{S['DKF']}:{S['DKF'] + 1}], {sp('REC')}, 16",
            f"s_add_u32 {s1('REC')}, {s1('REC')}, {s1('RECSTEP')}",
            f"s_addc_u32 {s1('REC', 1)}, {s1('REC', 1)}, {s1('RECSTEP', 1)}"]


DPP = [("quad_perm:[1,0,3,2]", "0xf"), ("quad_perm:[2,3,0,1]", "0xf"), ("row_half_mirror", "0xf"), ("row_mirror", "0xf")]


def reduction(G):
    x, t = V["X"], V["T"]
    o = [f"v_add_f64 {vp('X')}, {vp('X')}, {vp('T')}"]   # the two accumulators
    for ctrl, rm in DPP:
        o += ["s_nop 1",
              f"v_mov_b32_dpp v{t}, v{x} {ctrl} row_mask:{rm} bank_mask:0xf",
              f"v_mov_b32_dpp v{t + 1}, v{x + 1} {ctrl} row_mask:{rm} bank_mask:0xf",
              f"v_add_f64 {vp('X')}, {vp('X')}, {vp('T')}"]
    # row sums down the wave: rows 1 and 3 take in rows 0 and 2, then row 3 takes in row 1; lane 63 has the sum
    o += ["s_nop 1",
          f"v_mov_b32_dpp v{t}, v{x} row_bcast:15 row_mask:0xa bank_mask:0xf",
          f"v_mov_b32_dpp v{t + 1}, v{x + 1} row_bcast:15 row_mask:0xa bank_mask:0xf",
          "s_mov_b32 exec_lo, 0xffff0000",
          "s_mov_b32 exec_hi, 0xffff0000",
          f"v_add_f64 {vp('X')}, {vp('X')}, {vp('T')}",
          "s_mov_b64 exec, -1",
          "s_nop 0",
          f"v_mov_b32_dpp v{t}, v{x} row_bcast:31 row_mask:0xc bank_mask:0xf",
          f"v_mov_b32_dpp v{t + 1}, v{x + 1} row_bcast:31 row_mask:0xc bank_mask:0xf",
          "s_mov_b32 exec_lo, 0",
          f"v_add_f64 {vp('X')}, {vp('X')}, {vp('T')}"]
    if G == 1:
        o += ["s_mov_b64 exec, -1",
              "s_nop 0",
              f"v_readlane_b32 vcc_lo, v{x}, 63",
              f"v_readlane_b32 vcc_hi, v{x + 1}, 63",
              "s_nop 1",
              f"v_mov_b64 {vp('S')}, vcc",
              f"s_mov_b32 {s1('SHI')}, vcc_hi"]
        return o
    # G > 1: lane 63 of every wave drops its sum into this site's LDS slot; one barrier; every lane reads all G
    o += ["s_mov_b32 exec_hi, 0x80000000",   # exec_lo is 0 already
          f"ds_write_b64 {v1('LDSW')}, {vp('X')}",
          "s_mov_b64 exec, -1",
          "s_waitcnt lgkmcnt(0)",
          "s_barrier",
          f"ds_read_b128 v[{x}:{x + 3}], {v1('LDSR')}"]
    if G >= 4:
        o += [f"ds_read_b128 v[{V['RX']}:{V['RX'] + 3}], {v1('LDSR')} offset:16"]
    o += ["s_waitcnt lgkmcnt(0)"]
    if G == 2:
        o += [f"v_add_f64 {vp('S')}, {vp('X')}, {vp('T')}"]
    else:
        o += [f"v_add_f64 {vp('X')}, {vp('X')}, {vp('T')}",
              f"v_add_f64 {vp('X')}, {vp('X')}, {vp('RX')}"]
        if G == 4:
            o += [f"v_add_f64 {vp('S')}, {vp('X')}, {vp('S')}"]
        else:
            o += [f"v_add_f64 {vp('X')}, {vp('X')}, {vp('S')}",
                  f"ds_read_b128 v[{V['RX']}:{V['RX'] + 3}], {v1('LDSR')} offset:32",
                  "s_waitcnt lgkmcnt(0)",
                  f"v_add_f64 {vp('X')}, {vp('X')}, {vp('RX')}",
                  f"v_add_f64 {vp('X')}, {vp('X')}, {vp('S')}",
                  f"ds_read_b128 v[{V['RX']}:{V['RX'] + 3}], {v1('LDSR')} offset:48",
                  "s_waitcnt lgkmcnt(0)",
                  f"v_add_f64 {vp('X')}, {vp('X')}, {vp('RX')}",
                  f"v_add_f64 {vp('S')}, {vp('X')}, {vp('S')}"]
    # the other slot next time
    o += [f"v_xor_b32 {v1('LDSW')}, {8 * G}, {v1('LDSW')}",
          f"v_xor_b32 {v1('LDSR')}, {8 * G}, {v1('LDSR')}",
          "s_nop 0",
          f"v_readfirstlane_b32 {s1('SHI')}, v{V['S'] + 1}",
          "s_nop 1"]
    return o


def loop_body(mpl, G):
    o = []
    o += ["s_cmp_lg_u32 %[resume], 0",
          "s_cbranch_scc1 .Lcont%=",
          ".Lsite%=:"]
    # every wave of a SIMD cycles its priority with its progress (see paint_loop.hpp)
    o += [f"s_and_b32 {s1('T2')}, {s1('LEFT')}, 31",
          f"s_cmp_lg_u32 {s1('T2')}, 0",
          "s_cbranch_scc1 .Lnoprio%=",
          f"s_lshr_b32 {s1('T2')}, {s1('LEFT')}, 5",
          f"s_add_u32 {s1('T2')}, {s1('T2')}, {s1('RANK')}",
          f"s_and_b32 {s1('T2')}, {s1('T2')}, 3",
          f"s_cmp_eq_u32 {s1('T2')}, 0", "s_cbranch_scc1 .Lp0%=",
          f"s_cmp_eq_u32 {s1('T2')}, 1", "s_cbranch_scc1 .Lp1%=",
          f"s_cmp_eq_u32 {s1('T2')}, 2", "s_cbranch_scc1 .Lp2%=",
          "s_setprio 3", "s_branch .Lnoprio%=",
          ".Lp0%=:", "s_setprio 0", "s_branch .Lnoprio%=",
          ".Lp1%=:", "s_setprio 1", "s_branch .Lnoprio%=",
          ".Lp2%=:", "s_setprio 2",
          ".Lnoprio%=:"]
    o += [f"v_mul_f64 {vp('RX')}, {vp('S')}, {sp('COEF')}"]
    # lanes without a state <- -rx (the sweep adds rx to every lane: they come out as exact zeros)
    o += [f"s_cmp_eq_u32 {s1('HASSL')}, 0",
          "s_cbranch_scc1 .Lnosl%=",
          f"v_xor_b32 v{V['T'] + 1}, 0x80000000, v{V['RX'] + 1}",
          f"s_mov_b64 exec, {sp('SELFM')}",
          f"s_set_gpr_idx_on {s1('SELFIDX')}, 8",
          "s_nop 1",
          f"v_mov_b32 v{A0}, v{V['RX']}",
          f"v_mov_b32 v{A0 + 1}, v{V['T'] + 1}",
          "s_set_gpr_idx_off",
          f"s_mov_b32 {s1('CNT')}, {s1('PHIDX')}",
          f"s_mov_b64 exec, {sp('PHM')}",
          f"s_cmp_ge_u32 {s1('CNT')}, {2 * mpl}",
          "s_cbranch_scc1 .Lsldone%=",
          ".Lslloop%=:",
          f"s_set_gpr_idx_on {s1('CNT')}, 8",
          "s_nop 1",
          f"v_mov_b32 v{A0}, v{V['RX']}",
          f"v_mov_b32 v{A0 + 1}, v{V['T'] + 1}",
          "s_set_gpr_idx_off",
          "s_mov_b64 exec, -1",
          f"s_add_u32 {s1('CNT')}, {s1('CNT')}, 2",
          f"s_cmp_lt_u32 {s1('CNT')}, {2 * mpl}",
          "s_cbranch_scc1 .Lslloop%=",
          ".Lsldone%=:",
          "s_mov_b64 exec, -1",
          ".Lnosl%=:"]
    o += [f"v_add_f64 {vp('LS')}, {vp('LS')}, {sp('NX')}",
          f"s_cmp_eq_u32 {s1('LEFT')}, 0",
          f"s_cselect_b64 {sp('SKM')}, {sp('SKEDGE')}, -1"]
    rr = rounds(mpl)
    for r, (w0, nw) in enumerate(rr):
        for i in range(w0, w0 + nw):
            o.append(f"v_add_f64 {a(i)}, {a(i)}, {vp('RX')}")
        o.append("s_waitcnt lgkmcnt(0)")
        if r == 0:
            # the record fetched during the previous site: gap terms of the step after this site, dk two steps on
            o += [f"s_mov_b64 {sp('COEF')}, {sp('RECBUF')}",
                  f"s_mov_b64 {sp('NX')}, {sp('RECBUF', 2)}",
                  f"s_cmp_eq_u32 {s1('DIR')}, 0",
                  f"s_cselect_b32 {s1('DK2')}, {s1('DKF')}, {s1('DKB')}"]
        for i in range(w0, w0 + nw):
            o.append(f"s_andn2_b64 exec, {sp('SKM')}, s[{S['BUF'] + 2 * (i - w0)}:{S['BUF'] + 2 * (i - w0) + 1}]")
            o.append(f"v_mul_f64 {a(i)}, {a(i)}, {vp('RHO')}")
        o.append("s_mov_b64 exec, -1")
        if r + 1 < len(rr):
            o += sload("RP", rr[r + 1][0], rr[r + 1][1])
        else:
            o += sload("RPNEXT", rr[0][0], rr[0][1])
            # the record of the site after the next one; the last site of a pass has none (its list ends there:
            # one record further lies outside the target's list, for the first or last target outside the buffer)
            o += [f"s_cmp_eq_u32 {s1('LEFT')}, 0",
                  "s_cbranch_scc1 .Lnorec%="]
            o += fetch_record()
            o += [".Lnorec%=:",
                  f"s_mul_hi_u32 {s1('TMP', 1)}, {s1('DK2')}, {s1('STRIDE')}",
                  f"s_mul_i32 {s1('TMP')}, {s1('DK2')}, {s1('STRIDE')}",
                  f"s_add_u32 {s1('TMP')}, {s1('TMP')}, {s1('ROWSW')}",
                  f"s_addc_u32 {s1('TMP', 1)}, {s1('TMP', 1)}, {s1('ROWSW', 1)}",
                  f"global_load_dword {v1('JUNK')}, {v1('LANEB')}, {sp('TMP')}"]
            if mpl > 32:
                o.append(f"global_load_dword {v1('JUNK')}, {v1('LANEB')}, {sp('TMP')} offset:256")
            if mpl > 64:
                o.append(f"global_load_dword {v1('JUNK')}, {v1('LANEB')}, {sp('TMP')} offset:512")
        for i in range(w0, w0 + nw):
            acc = "X" if i % 2 == 0 else "T"
            if i == 0 or i == 1:
                continue
            if i == 2:
                o.append(f"v_add_f64 {vp('X')}, {a(0)}, {a(2)}")
            elif i == 3:
                o.append(f"v_add_f64 {vp('T')}, {a(1)}, {a(3)}")
            else:
                o.append(f"v_add_f64 {vp(acc)}, {vp(acc)}, {a(i)}")
        if w0 + nw <= 3:  # a first round of fewer than four words cannot happen with B >= 4 and MPL >= 4
            raise ValueError("round too small")
    o += reduction(G)
    o += [f"s_sub_u32 {s1('T2')}, {s1('SHI')}, 0x3de00000",
          f"s_cmp_lt_u32 {s1('T2')}, 0x04200000",       # [2^-33, 2^33): surely inside the rescale thresholds
          "s_cbranch_scc0 .Lexit%=",
          f"s_cmp_eq_u32 {s1('LEFT')}, {s1('STONELEFT')}",
          "s_cbranch_scc1 .Lexit%=",
          ".Lcont%=:",
          f"s_mov_b64 {sp('RP')}, {sp('RPNEXT')}",
          f"s_mov_b64 {sp('RPNEXT')}, {sp('TMP')}",
          f"s_sub_u32 {s1('LEFT')}, {s1('LEFT')}, 1",     # borrow (SCC) once the last site has been done
          "s_cbranch_scc0 .Lsite%=",
          ".Lexit%=:",
          f"s_mov_b32 %[left], {s1('LEFT')}",
          f"s_mov_b32 %[shi], {s1('SHI')}"]
    return o


def clobbers(mpl):
    v = [f'"v{i}"' for i in range(CV, A0 + 2 * mpl)]
    s = [f'"s{i}"' for i in range(CS, 102)]
    return ", ".join(v + s + ['"vcc"', '"scc"', '"memory"'])


def main():
    p = print
    p("// GENERATED by tools/gen_paint_loop.py -- do not edit; the generator is the source (its header explains the loop).")
    p("#pragma once")
    p("#include <hip/hip_runtime.h>")
    p("#include <stdint.h>")
    p("namespace relate {")
    p("namespace aloop {")
    p(f"constexpr int kCV = {CV}, kCS = {CS}, kRowWords = {B}, kA0 = {A0};")
    for k, val in V.items():
        p(f"constexpr int kV_{k} = {val};")
    for k, val in S.items():
        p(f"constexpr int kS_{k} = {val};")
    p("template <int MPL, int G> struct HotLoop;")
    p("template <int MPL> struct Claim;")
    p("// Row words of one round (at most kRowWords words) into s[kS_BUF ..]: asynchronous until s_waitcnt lgkmcnt(0).")
    p("template <int MPL> struct Rows;")
    for mpl in MPLS:
        p(f"template <> struct Claim<{mpl}> {{ static __device__ __forceinline__ void all() {{ asm volatile(\"\" ::: {clobbers(mpl)}); }} }};")
        rr = rounds(mpl)
        p(f"template <> struct Rows<{mpl}> {{")
        p(f"  static constexpr int NR = {len(rr)};")
        p("  static constexpr int first(int r) { constexpr int f[] = {" + ", ".join(str(f) for f, _ in rr) + "}; return f[r]; }")
        p("  static constexpr int size(int r) { constexpr int n[] = {" + ", ".join(str(n) for _, n in rr) + "}; return n[r]; }")
        p("  // round R of the current row (s[kS_RP])")
        p("  template <int R> static __device__ __forceinline__ void of_current() {")
        for r, (f, n) in enumerate(rr):
            body = "\\n\\t".join(sload("RP", f, n))
            p(f"    if constexpr (R == {r}) asm volatile(\"{body}\" ::: \"memory\");")
        p("  }")
        body = "\\n\\t".join(sload("RPNEXT", rr[0][0], rr[0][1]))
        p("  // first round of the next row (s[kS_RPNEXT]): what the loop expects to be on its way when it is entered")
        p(f"  static __device__ __forceinline__ void first_of_next() {{ asm volatile(\"{body}\" ::: \"memory\"); }}")
        p("};")
        for G in GS:
            body = loop_body(mpl, G)
            p(f"template <> struct HotLoop<{mpl}, {G}> {{")
            p("  // returns the sites still to come (-1: the list is finished); shi = high word of the sum")
            p("  static __device__ __forceinline__ int run(uint32_t resume, uint32_t& shi) {")
            p("    int left;")
            p("    asm volatile(")
            for line in body:
                p(f'        "{line}\\n\\t"')
            p("        : [left] \"=s\"(left), [shi] \"=s\"(shi)")
            p("        : [resume] \"s\"(resume)")
            p(f"        : {clobbers(mpl)});")
            p("    return left;")
            p("  }")
            p("};")
    p("}  // namespace aloop")
    p("}  // namespace relate")


if __name__ == "__main__":
    main()
