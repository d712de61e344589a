// Register-resident gfx950 recursion kernel of the stepping-stone painting
// (forward and backward Li-Stephens passes of one target per workgroup).
//
// The states of a wave (MPL doubles per lane) live in a fixed block of VGPRs that the compiler never
// sees: the kernel is compiled with amdgpu_num_vgpr / amdgpu_num_sgpr, which keeps the register
// allocator below v[kCV] / s[kCS], and everything above is addressed by number from inline asm.
// Hence no spills, no copies at control-flow joins, and the sweep over the states is straight-line
// code:
//   forward   a' = (a + rx) * (mismatch ? rho : 1)           add, masked mul, accumulate  (3 per 64 states)
//   backward  the same recursion on g = beta * emission when painter and data hold one theta; otherwise
//             t = b + add_all; b' = mismatch' ? fma(t, rho, c1) : t   add, masked fma,
//             sums of b' over all / over mismatching states  accumulate, masked accumulate (4 per 64 states)
// "masked" = the bit-row word of the site (an SGPR pair fed by s_load_dwordx16 / x8) written into EXEC,
// so the emission costs no vector instruction.
//
// Lanes without a state (the target itself, lanes past N) never mismatch -- the target carries the
// derived allele at its own sites and pack_rows sets the bits past N -- so their update is "+ rx":
// before every sweep they are loaded with -rx (VGPR index mode reaches the register of the target's
// own word) and come out as exact zeros, which keeps every sum and every stone free of masks.
//
// gfx950 data hazards are invisible to the compiler inside an asm statement; the statements below
// carry their own wait states (s_nop) where a vector result feeds v_readlane / DPP / an SGPR read.
//
// Internal header; instantiated per G by paint_regs_g*.hip.
#pragma once
#include <type_traits>

#include "paint_device.hpp"

namespace relate {
namespace regs {

using namespace dev;

// ---- register map -------------------------------------------------------------------------------
#define RELATE_CV 48  // compiler: v0..v47
#ifndef RELATE_CS
#define RELATE_CS 52  // compiler: s0..s51 (44: s0..s43, and 20 instead of 16 row words per round)
#endif
// amdgpu_num_sgpr counts the 8 special registers (VCC, FLAT_SCRATCH, XNACK_MASK) on top of the compiler's
#define RELATE_CS_ATTR (RELATE_CS + 8)
// amdgpu_num_vgpr is doubled by the gfx90a+ back end (unified VGPR/AGPR file); no AGPRs are used here
#define RELATE_CV_ATTR 24
constexpr int kCV = RELATE_CV, kCS = RELATE_CS;
static_assert((RELATE_CS == 52 || RELATE_CS == 44) && 2 * RELATE_CV_ATTR == RELATE_CV, "register limits");
constexpr int kX = kCV;       // v[kX:kX+1]   reduction operand
constexpr int kT = kCV + 2;   // v[kT:kT+1]   scratch pair
constexpr int kLo = kCV + 4;  // v[kLo:kLo+1] lower rescale threshold, v[kHi:kHi+1] upper
constexpr int kHi = kCV + 6;
constexpr int kA0 = kCV + 8;  // v[kA0 + 2i : kA0 + 2i + 1] = state of word i
// Values that every site of a pass reads and none writes: held here, the compiler cannot decide to park
// them in VGPR lanes and fetch them back (v_readlane, a vector instruction) at every site.
constexpr int kRowsW = kCS;        // s[kRowsW:kRowsW+1] first row word of this wave (row 0)
constexpr int kSelfIdx = kCS + 2;  // 2 * local word of the target's own lane
constexpr int kPhIdx = kCS + 3;    // 2 * local word of the first padding lane (2*MPL: none)
constexpr int kSelfM = kCS + 4;    // s[kSelfM:kSelfM+1] the target's own lane (0: another wave has it)
constexpr int kPhM = kCS + 6;      // s[kPhM:kPhM+1] padding lanes of word kPhIdx/2
// s[kBufA .. 91]: row words, kBufWords of them; the two-row sweeps use them as two halves of 16 registers
constexpr int kBufA = kCS + 8;
constexpr int kBufWords = (92 - kBufA) / 2;
constexpr int kBufWords2 = kBufWords / 2;     // per row where a round holds the words of two rows
constexpr int kBufC = kBufA + 2 * kBufWords2;  // the second row's words
constexpr int kCoefN = 92, kNxN = 94;  // prefetched per-gap terms (pairs; one 4-aligned quad)
constexpr int kDkN = 96;  // prefetched list entry
constexpr int kCnt = 97;  // loop counter of the padding-lane writes
constexpr int kRec = 98;  // s[kRec:kRec+1] running pointer into the site records
constexpr int kSkEdge = 100;  // s[kSkEdge:kSkEdge+1] the target's own allele at the forced end site of the pass, as a mask
constexpr int kMaxMPL = (256 - kA0) / 2;
static_assert(kSkEdge + 1 <= 101 && kCoefN % 4 == 0 && kBufA % 4 == 0, "SGPR map");

#define RELATE_REGS_ATTR __attribute__((amdgpu_num_vgpr(RELATE_CV_ATTR), amdgpu_num_sgpr(RELATE_CS_ATTR)))

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Tells the compiler which registers the asm statements use by number: it then counts them in the
// kernel's register budget and keeps its own SGPR-spill lanes out of them (the spill VGPRs are taken
// from the registers it believes unused, whatever amdgpu_num_vgpr says).  Clobber names must be
// literals, hence the cumulative lists: RELATE_VCLOB_<MPL> = v48 .. v[kA0 + 2*MPL - 1], RELATE_SCLOB = s52 .. s101.
#define RELATE_VCLOB_4 "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
#define RELATE_VCLOB_8 RELATE_VCLOB_4, "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"
#define RELATE_VCLOB_12 RELATE_VCLOB_8, "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"
#define RELATE_VCLOB_16 RELATE_VCLOB_12, "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87"
#define RELATE_VCLOB_20 RELATE_VCLOB_16, "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95"
#define RELATE_VCLOB_24 RELATE_VCLOB_20, "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103"
#define RELATE_VCLOB_28 RELATE_VCLOB_24, "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111"
#define RELATE_VCLOB_32 RELATE_VCLOB_28, "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119"
#define RELATE_VCLOB_36 RELATE_VCLOB_32, "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
#define RELATE_VCLOB_40 RELATE_VCLOB_36, "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135"
#define RELATE_VCLOB_48 RELATE_VCLOB_40, "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151"
#define RELATE_VCLOB_56 RELATE_VCLOB_48, "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"
#define RELATE_VCLOB_64 RELATE_VCLOB_56, "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183"
#define RELATE_VCLOB_72 RELATE_VCLOB_64, "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199"
#define RELATE_VCLOB_80 RELATE_VCLOB_72, "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215"
#if RELATE_CS == 44
#define RELATE_SCLOB_LOW "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
#else
#define RELATE_SCLOB_LOW
#endif
#define RELATE_SCLOB RELATE_SCLOB_LOW "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101"
template <int MPL>
__device__ __forceinline__ void claim_registers();
#define RELATE_CLAIM(M)                                                \
  template <>                                                          \
  __device__ __forceinline__ void claim_registers<M>()                 \
  {                                                                    \
    asm volatile("; relate_limits v%0 s%1" ::"n"(RELATE_CV), "n"(RELATE_CS) : RELATE_VCLOB_##M, RELATE_SCLOB); \
  }
RELATE_CLAIM(4)
RELATE_CLAIM(8)
RELATE_CLAIM(12)
RELATE_CLAIM(16)
RELATE_CLAIM(20)
RELATE_CLAIM(24)
RELATE_CLAIM(28)
RELATE_CLAIM(32)
RELATE_CLAIM(36)
RELATE_CLAIM(40)
RELATE_CLAIM(48)
RELATE_CLAIM(56)
RELATE_CLAIM(64)
RELATE_CLAIM(72)
RELATE_CLAIM(80)
#undef RELATE_CLAIM
static_assert(kCV == 48 && kA0 == 56 && (kCS == 52 || kCS == 44), "the clobber lists above are written for this register map (tools/gen_clobbers.py 48 56 52)");

// A wave-uniform value as an SGPR (the compiler's readfirstlane may be folded into a VGPR copy).
__device__ __forceinline__ uint32_t sgpr32(uint32_t x)
{
  uint32_t r;
  asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ uint64_t sgpr64(uint64_t x)
{
  const uint32_t lo = sgpr32((uint32_t)x), hi = sgpr32((uint32_t)(x >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double sgpr_f64(double x) { return __longlong_as_double((long long)sgpr64((uint64_t)__double_as_longlong(x))); }
// (j == j_edge) ? edge_mask : all ones, on the scalar unit.
__device__ __forceinline__ uint64_t edge_select(int j, int j_edge, uint64_t edge_mask)
{
  uint64_t r;
  asm volatile("s_cmp_eq_u32 %1, %2\n\ts_cselect_b64 %0, %3, -1" : "=s"(r) : "s"(j), "s"(j_edge), "s"(edge_mask) : "scc");
  return r;
}

// The same with the mask held in s[kSkEdge:kSkEdge+1] (set_edge_mask).
__device__ __forceinline__ void set_edge_mask(uint64_t m) { asm volatile("s_mov_b64 s[%1:%1+1], %0" ::"s"(m), "n"(kSkEdge)); }
__device__ __forceinline__ uint64_t edge_mask(int j, int j_edge)
{
  uint64_t r;
  asm volatile("s_cmp_eq_u32 %1, %2\n\ts_cselect_b64 %0, s[%3:%3+1], -1" : "=s"(r) : "s"(j), "s"(j_edge), "n"(kSkEdge) : "scc");
  return r;
}
// First row word of this wave in row `site`: s[kRowsW:kRowsW+1] + site * stride_bytes (set_rows_base).
__device__ __forceinline__ void set_rows_base(const uint64_t* p) { asm volatile("s_mov_b64 s[%1:%1+1], %0" ::"s"(p), "n"(kRowsW)); }
__device__ __forceinline__ const uint64_t* row_pointer(int site, uint32_t stride_bytes)
{
  uint32_t lo, hi;
  asm volatile(
      "s_mul_hi_u32 %1, %2, %3\n\t"
      "s_mul_i32 %0, %2, %3\n\t"
      "s_add_u32 %0, %0, s[%4]\n\t"
      "s_addc_u32 %1, %1, s[%4+1]"
      : "=&s"(lo), "=&s"(hi)
      : "s"(site), "s"(stride_bytes), "n"(kRowsW)
      : "scc");
  return reinterpret_cast<const uint64_t*>(((uint64_t)hi << 32) | lo);
}

// ---- row words -------------------------------------------------------------------------------------
// 4 row words (32 bytes) at p + OFF bytes -> s[SB:SB+7]; asynchronous until swait().
template <int SB, int OFF>
__device__ __forceinline__ void sload4(const uint64_t* p)
{
  asm volatile("s_load_dwordx8 s[%1:%1+7], %0, %2" ::"s"(p), "n"(SB), "n"(OFF) : "memory");
}
__device__ __forceinline__ void swait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// NW (1..8) row words at p + OFF bytes -> s[SB..]: five or more travel as one 16-dword load.  The load
// may fetch up to 3 words past the ones asked for (the packed rows are allocated with that slack).
template <int SB, int OFF, int NW>
__device__ __forceinline__ void sload_words(const uint64_t* p)
{
  static_assert(NW >= 1 && NW <= 8 && SB >= kBufA && SB % 4 == 0 && SB + (NW > 4 ? 16 : 8) <= 92, "a part of the row-word registers");
  if constexpr (NW > 4)
    asm volatile("s_load_dwordx16 s[%1:%1+15], %0, %2" ::"s"(p), "n"(SB), "n"(OFF) : "memory");
  else
    sload4<SB, OFF>(p);
}
// The MPL words of a wave in rounds of at most 16 words, the capacity of s[kBufA .. kBufA+31].  Scalar loads
// return out of order, so the only wait there is waits for all of them, and the time available to hide a
// fetch is what lies between its issue and the next wait.  A round therefore runs: additions of its words
// (no masks) -- wait -- masked multiplications (the last use of the masks) -- fetch of the next round --
// accumulation of its words (no masks): every fetch has two thirds of a round to arrive, the fetch for the
// next site also the reduction, and a site takes as many waits as it has rounds.
template <int MPL>
struct RowRounds {
  static constexpr int NR = (MPL + kBufWords - 1) / kBufWords;
  static constexpr int size(int r) { return MPL / NR + (r < MPL % NR ? 1 : 0); }
  static constexpr int first(int r)
  {
    int o = 0;
    for (int i = 0; i < r; i++) o += size(i);
    return o;
  }
};
// words [W0, W0 + NW) of the row at p -> s[kBufA ..], word W0 + i in s[kBufA + 2i : kBufA + 2i + 1]
template <int W0, int NW>
__device__ __forceinline__ void sload_round(const uint64_t* p)
{
  static_assert(NW >= 1 && NW <= kBufWords, "one round");
  constexpr int na = NW < 8 ? NW : 8;
  sload_words<kBufA, 8 * W0, na>(p);
  if constexpr (NW > 8) sload_words<kBufA + 16, 8 * (W0 + 8), (NW - 8 < 8 ? NW - 8 : 8)>(p);
  if constexpr (NW > 16) sload_words<kBufA + 32, 8 * (W0 + 16), NW - 16>(p);
}
// The record of a list entry (SiteRec: coef, nx, dk_fwd2, dk_bwd2), fetched during a sweep for the site
// after it.  The two list entries land in s[kDkN] (forward) and s[kCnt] (backward); the latter doubles
// as the loop counter of write_stateless(), which has finished before the fetch is issued.
template <bool FWD>
__device__ __forceinline__ void prefetch_site()
{
  // through the running pointer s[kRec:kRec+1], which then moves on to the next record of the pass
  if constexpr (FWD)
    asm volatile(
        "s_load_dwordx4 s[%0:%0+3], s[%2:%2+1], 0\n\t"
        "s_load_dwordx2 s[%1:%1+1], s[%2:%2+1], 16\n\t"
        "s_add_u32 s[%2], s[%2], 24\n\t"
        "s_addc_u32 s[%2+1], s[%2+1], 0" ::"n"(kCoefN),
        "n"(kDkN), "n"(kRec)
        : "memory", "scc");
  else
    asm volatile(
        "s_load_dwordx4 s[%0:%0+3], s[%2:%2+1], 0\n\t"
        "s_load_dwordx2 s[%1:%1+1], s[%2:%2+1], 16\n\t"
        "s_sub_u32 s[%2], s[%2], 24\n\t"
        "s_subb_u32 s[%2+1], s[%2+1], 0" ::"n"(kCoefN),
        "n"(kDkN), "n"(kRec)
        : "memory", "scc");
}
__device__ __forceinline__ void set_site_pointer(const SiteRec* q)
{
  asm volatile("s_mov_b64 s[%1:%1+1], %0" ::"s"(q), "n"(kRec));
}
// Rescale thresholds into their registers; out_of_range(S) = S < lo || S > hi as a lane mask (all or none:
// S is the same in every lane).
__device__ __forceinline__ void set_thresholds(double lo, double hi)
{
  asm volatile("v_mov_b64 v[%2:%2+1], %0\n\tv_mov_b64 v[%3:%3+1], %1" ::"v"(lo), "v"(hi), "n"(kLo), "n"(kHi));
}
__device__ __forceinline__ bool out_of_range(double S)
{
  uint64_t m;
  asm volatile(
      "v_cmp_lt_f64 vcc, %1, v[%2:%2+1]\n\t"
      "v_cmp_gt_f64 %0, %1, v[%3:%3+1]\n\t"
      "s_or_b64 %0, %0, vcc"
      : "=s"(m)
      : "v"(S), "n"(kLo), "n"(kHi)
      : "vcc", "scc");
  return m != 0;
}
// Rows come through the scalar cache from L2; a row that has left L2 costs the wave several times the usual
// wait.  A vector load of the row (MPL words from p on: one dword per lane covers 32 words) issued two sites
// ahead pulls it back in time.  The data lands in v[kJunk], which nothing reads; loops that use this keep
// their rescale thresholds out of v[kLo..kHi+1] (out_of_range_exact).
constexpr int kJunk = kLo;
constexpr int kLaneBytes = kLo + 1;  // 4 * lane, the address offset of those loads (set_touch_lane)
constexpr bool kTouchRows = true;
__device__ __forceinline__ void set_touch_lane(uint32_t lane_bytes) { asm volatile("v_mov_b32 v[%1], %0" ::"v"(lane_bytes), "n"(kLaneBytes)); }
template <int MPL>
__device__ __forceinline__ void touch_row(const uint64_t* p)
{
  asm volatile("global_load_dword v[%1], v[%2], %0" ::"s"(p), "n"(kJunk), "n"(kLaneBytes) : "memory");
  if constexpr (MPL > 32) asm volatile("global_load_dword v[%1], v[%2], %0 offset:256" ::"s"(p), "n"(kJunk), "n"(kLaneBytes) : "memory");
  if constexpr (MPL > 64) asm volatile("global_load_dword v[%1], v[%2], %0 offset:512" ::"s"(p), "n"(kJunk), "n"(kLaneBytes) : "memory");
}
// A sum in [2^-33, 2^33) lies strictly inside the rescale thresholds (kRescaleLo = 1e-10 and its reciprocal):
// one scalar compare on the high word settles nearly every site; the rest (NaN and negative values included)
// takes the exact comparison.
static_assert(kRescaleLo < 0x1p-33 && 1.0 / kRescaleLo > 0x1p33, "band of the scalar range check");
__device__ __forceinline__ bool surely_in_range(uint32_t high_word) { return high_word - 0x3DE00000u < 0x42000000u - 0x3DE00000u; }
template <bool FWD>
__device__ __forceinline__ void fetched_site(int& s, double& coef, double& nx)
{
  // s_mov into compiler registers: from here on they are ordinary values
  uint32_t lo0, hi0, lo1, hi1;
  asm volatile(
      "s_mov_b32 %0, s[%5]\n\ts_mov_b32 %1, s[%6]\n\ts_mov_b32 %2, s[%6+1]\n\ts_mov_b32 %3, s[%7]\n\t"
      "s_mov_b32 %4, s[%7+1]"
      : "=s"(s), "=s"(lo0), "=s"(hi0), "=s"(lo1), "=s"(hi1)
      : "n"(FWD ? kDkN : kCnt), "n"(kCoefN), "n"(kNxN));
  coef = __hiloint2double((int)hi0, (int)lo0);
  nx = __hiloint2double((int)hi1, (int)lo1);
}
static_assert(kCnt == kDkN + 1 && kNxN == kCoefN + 2 && sizeof(SiteRec) == 24, "record layout against the SGPR map");
// The same values used where they arrive (the loops of several waves per target): no copies into compiler
// registers, which are scarce there.  x * coef, ls += nx, and the row of the fetched list entry.
__device__ __forceinline__ void set_site_terms(double coef, double nx)
{
  asm volatile("s_mov_b64 s[%2:%2+1], %0\n\ts_mov_b64 s[%3:%3+1], %1" ::"s"(coef), "s"(nx), "n"(kCoefN), "n"(kNxN));
}
__device__ __forceinline__ double times_site_coef(double x)
{
  double r;
  asm volatile("v_mul_f64 %0, %1, s[%2:%2+1]" : "=v"(r) : "v"(x), "n"(kCoefN));
  return r;
}
__device__ __forceinline__ void add_site_nx(double& ls) { asm volatile("v_add_f64 %0, %0, s[%1:%1+1]" : "+v"(ls) : "n"(kNxN)); }
template <bool FWD>
__device__ __forceinline__ const uint64_t* fetched_row_pointer(uint32_t stride_bytes)
{
  uint32_t lo, hi;
  asm volatile(
      "s_mul_hi_u32 %1, s[%2], %3\n\t"
      "s_mul_i32 %0, s[%2], %3\n\t"
      "s_add_u32 %0, %0, s[%4]\n\t"
      "s_addc_u32 %1, %1, s[%4+1]"
      : "=&s"(lo), "=&s"(hi)
      : "n"(FWD ? kDkN : kCnt), "s"(stride_bytes), "n"(kRowsW)
      : "scc");
  return reinterpret_cast<const uint64_t*>(((uint64_t)hi << 32) | lo);
}

// ---- state block -----------------------------------------------------------------------------------
template <int I>
__device__ __forceinline__ void state_set(double v)
{
  asm volatile("v_mov_b64 v[%1:%1+1], %0" ::"v"(v), "n"(kA0 + 2 * I));
}
template <int I>
__device__ __forceinline__ void state_scale(double f)
{
  asm volatile("v_mul_f64 v[%1:%1+1], v[%1:%1+1], %0" ::"v"(f), "n"(kA0 + 2 * I));
}
// float(state I) -> LDS at lds_addr + 256*I (one dword per lane)
template <int I>
__device__ __forceinline__ void state_to_lds(uint32_t lds_addr)
{
  asm volatile("v_cvt_f32_f64 v[%2], v[%1:%1+1]\n\tds_write_b32 %0, v[%2] offset:%3" ::"v"(lds_addr), "n"(kA0 + 2 * I),
               "n"(kT + (I & 1)), "n"(256 * I)
               : "memory");
}

// float(state I * (inv_mis where the lane mismatches row word s[SW:SW+1] under sk, else inv_match)) -> LDS
template <int I, int SW>
__device__ __forceinline__ void state_unemitted_to_lds(uint32_t lds_addr, double inv_match, double inv_mis, uint64_t sk)
{
  asm volatile(
      "v_mul_f64 v[%[t]:%[t]+1], v[%[a]:%[a]+1], %[im]\n\t"
      "s_andn2_b64 exec, %[sk], s[%[w]:%[w]+1]\n\t"
      "v_mul_f64 v[%[t]:%[t]+1], v[%[a]:%[a]+1], %[ix]\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_cvt_f32_f64 v[%[x]], v[%[t]:%[t]+1]\n\t"
      "ds_write_b32 %[addr], v[%[x]] offset:%[off]" ::[addr] "v"(lds_addr),
      [im] "v"(inv_match), [ix] "v"(inv_mis), [sk] "s"(sk), [a] "n"(kA0 + 2 * I), [t] "n"(kT), [x] "n"(kX + (I & 1)),
      [w] "n"(SW), [off] "n"(256 * I)
      : "memory", "scc");
}

// Lanes without a state <- -(x*y): the target's own lane (s[kSelfM..] = its bit, or 0 if its word belongs to
// another wave; s[kSelfIdx] = 2 * local word) and every lane from the first padding lane of the wave on
// (s[kPhM..] = lanes of that word, s[kPhIdx] = 2 * its local word; 2*MPL if the wave has none).  x*y is what the
// sweep adds to these lanes (the product rounds like the caller's own x*y, so the sum is an exact zero).
// Bit 31 of s[kSelfIdx] says that the wave has no such lane at all (`end_idx` = 2*MPL): write_stateless_site then
// branches over its vector instructions.
__device__ __forceinline__ void set_stateless(uint64_t selfm, uint32_t selfidx, uint64_t phm, uint32_t phidx, uint32_t end_idx)
{
  if (selfm == 0 && phidx >= end_idx) selfidx |= 0x80000000u;
  asm volatile(
      "s_mov_b64 s[%4:%4+1], %0\n\ts_mov_b32 s[%5], %1\n\ts_mov_b64 s[%6:%6+1], %2\n\ts_mov_b32 s[%7], %3" ::"s"(selfm),
      "s"(selfidx), "s"(phm), "s"(phidx), "n"(kSelfM), "n"(kSelfIdx), "n"(kPhM), "n"(kPhIdx));
}
#define RELATE_WRITE_STATELESS_HEAD                       \
  "s_bitcmp1_b32 s[%[selfidx]], 31\n\t"                   \
  "s_cbranch_scc1 .Lws_done%=\n\t"
#define RELATE_WRITE_STATELESS_BODY                       \
  "s_mov_b64 exec, s[%[selfm]:%[selfm]+1]\n\t"            \
  "s_set_gpr_idx_on s[%[selfidx]], 8\n\t"                 \
  "s_nop 1\n\t"                                           \
  "v_mov_b32 v[%[a0]], v[%[t]]\n\t"                       \
  "v_mov_b32 v[%[a0]+1], v[%[t]+1]\n\t"                   \
  "s_set_gpr_idx_off\n\t"                                 \
  "s_mov_b32 s[%[cnt]], s[%[phidx]]\n\t"                  \
  "s_mov_b64 exec, s[%[phm]:%[phm]+1]\n\t"                \
  "s_cmp_ge_u32 s[%[cnt]], %[end]\n\t"                    \
  "s_cbranch_scc1 .Lws_done%=\n"                          \
  ".Lws_loop%=:\n\t"                                      \
  "s_set_gpr_idx_on s[%[cnt]], 8\n\t"                     \
  "s_nop 1\n\t"                                           \
  "v_mov_b32 v[%[a0]], v[%[t]]\n\t"                       \
  "v_mov_b32 v[%[a0]+1], v[%[t]+1]\n\t"                   \
  "s_set_gpr_idx_off\n\t"                                 \
  "s_mov_b64 exec, -1\n\t"                                \
  "s_add_u32 s[%[cnt]], s[%[cnt]], 2\n\t"                 \
  "s_cmp_lt_u32 s[%[cnt]], %[end]\n\t"                    \
  "s_cbranch_scc1 .Lws_loop%=\n"                          \
  ".Lws_done%=:\n\t"                                      \
  "s_mov_b64 exec, -1"
#define RELATE_WRITE_STATELESS_REGS                                                                              \
  [selfm] "n"(kSelfM), [selfidx] "n"(kSelfIdx), [phm] "n"(kPhM), [phidx] "n"(kPhIdx), [t] "n"(kT), [a0] "n"(kA0), \
      [cnt] "n"(kCnt), [end] "n"(2 * MPL)
// YS: y is wave-uniform and comes from a scalar register (no copy into a vector register)
template <int MPL, bool YS>
__device__ __forceinline__ void write_stateless(double x, double y)
{
  if constexpr (YS)
    asm volatile("v_mul_f64 v[%[t]:%[t]+1], %[x], -%[y]\n\t" RELATE_WRITE_STATELESS_BODY
                 :
                 : [x] "v"(x), [y] "s"(y), RELATE_WRITE_STATELESS_REGS
                 : "scc", "memory");
  else
    asm volatile("v_mul_f64 v[%[t]:%[t]+1], %[x], -%[y]\n\t" RELATE_WRITE_STATELESS_BODY
                 :
                 : [x] "v"(x), [y] "v"(y), RELATE_WRITE_STATELESS_REGS
                 : "scc", "memory");
}
// y = the fetched coefficient where it arrived (times_site_coef); a wave without such lanes (bit 31 of s[kSelfIdx],
// see set_stateless) branches over the vector instructions
template <int MPL>
__device__ __forceinline__ void write_stateless_site(double x)
{
  asm volatile(RELATE_WRITE_STATELESS_HEAD "v_mul_f64 v[%[t]:%[t]+1], %[x], -s[%[cf]:%[cf]+1]\n\t" RELATE_WRITE_STATELESS_BODY
               :
               : [x] "v"(x), [cf] "n"(kCoefN), RELATE_WRITE_STATELESS_REGS
               : "scc", "memory");
}
#undef RELATE_WRITE_STATELESS_HEAD
#undef RELATE_WRITE_STATELESS_BODY
#undef RELATE_WRITE_STATELESS_REGS

// ---- sweeps ----------------------------------------------------------------------------------------
#define RELATE_A(i) "v[%[r]+" #i "*2:%[r]+" #i "*2+1]"
#define RELATE_W(sb, i) "s[%[" #sb "]+" #i "*2:%[" #sb "]+" #i "*2+1]"

#define RELATE_FWD_ADD(i) "v_add_f64 " RELATE_A(i) ", " RELATE_A(i) ", %[rx]\n\t"
#define RELATE_FWD_MUL(i) "s_andn2_b64 exec, %[sk], " RELATE_W(s, i) "\n\tv_mul_f64 " RELATE_A(i) ", " RELATE_A(i) ", %[rho]\n\t"
#define RELATE_FWD_ACC(c, i) "v_add_f64 %[" #c "], %[" #c "], " RELATE_A(i) "\n\t"
#define RELATE_FWD_SET(c, i) "v_mov_b64 %[" #c "], " RELATE_A(i) "\n\t"
#define RELATE_FWD_SUM2(c, i, k) "v_add_f64 %[" #c "], " RELATE_A(i) ", " RELATE_A(k) "\n\t"

// The three kinds of instruction of the one-row recursion on up to 4 words from state word W0 on.
template <int W0, int NW>
__device__ __forceinline__ void row_add(double rx)
{
#define RELATE_OPS : : [rx] "v"(rx), [r] "n"(kA0 + 2 * W0)
  if constexpr (NW == 4) asm volatile(RELATE_FWD_ADD(0) RELATE_FWD_ADD(1) RELATE_FWD_ADD(2) RELATE_FWD_ADD(3) RELATE_OPS);
  else if constexpr (NW == 3) asm volatile(RELATE_FWD_ADD(0) RELATE_FWD_ADD(1) RELATE_FWD_ADD(2) RELATE_OPS);
  else if constexpr (NW == 2) asm volatile(RELATE_FWD_ADD(0) RELATE_FWD_ADD(1) RELATE_OPS);
  else asm volatile(RELATE_FWD_ADD(0) RELATE_OPS);
#undef RELATE_OPS
}
// masks in s[SB ..]
template <int W0, int SB, int NW>
__device__ __forceinline__ void row_mul(double rho, uint64_t sk)
{
#define RELATE_OPS : : [rho] "v"(rho), [sk] "s"(sk), [r] "n"(kA0 + 2 * W0), [s] "n"(SB) : "scc"
  if constexpr (NW == 4)
    asm volatile(RELATE_FWD_MUL(0) RELATE_FWD_MUL(1) RELATE_FWD_MUL(2) RELATE_FWD_MUL(3) "s_mov_b64 exec, -1" RELATE_OPS);
  else if constexpr (NW == 3) asm volatile(RELATE_FWD_MUL(0) RELATE_FWD_MUL(1) RELATE_FWD_MUL(2) "s_mov_b64 exec, -1" RELATE_OPS);
  else if constexpr (NW == 2) asm volatile(RELATE_FWD_MUL(0) RELATE_FWD_MUL(1) "s_mov_b64 exec, -1" RELATE_OPS);
  else asm volatile(RELATE_FWD_MUL(0) "s_mov_b64 exec, -1" RELATE_OPS);
#undef RELATE_OPS
}
// sums into two accumulators, which the first words of a site initialise (FIRST)
template <int W0, int NW, bool FIRST>
__device__ __forceinline__ void row_acc(double& c0, double& c1)
{
  static_assert(!FIRST || NW >= 2, "the first words initialise both accumulators");
#define RELATE_OPS : [c0] "+v"(c0), [c1] "+v"(c1) : [r] "n"(kA0 + 2 * W0)
  // the first words only write the accumulators: as outputs, so that what the variables held before (nothing, at the
  // first site of a loop) is not a value the compiler carries round the loop
#define RELATE_OPS_FIRST : [c0] "=v"(c0), [c1] "=v"(c1) : [r] "n"(kA0 + 2 * W0)
  if constexpr (NW == 4 && FIRST) asm volatile(RELATE_FWD_SUM2(c0, 0, 2) RELATE_FWD_SUM2(c1, 1, 3) RELATE_OPS_FIRST);
  else if constexpr (NW == 4) asm volatile(RELATE_FWD_ACC(c0, 0) RELATE_FWD_ACC(c1, 1) RELATE_FWD_ACC(c0, 2) RELATE_FWD_ACC(c1, 3) RELATE_OPS);
  else if constexpr (NW == 3 && FIRST) asm volatile(RELATE_FWD_SUM2(c0, 0, 2) RELATE_FWD_SET(c1, 1) RELATE_OPS_FIRST);
  else if constexpr (NW == 3) asm volatile(RELATE_FWD_ACC(c0, 0) RELATE_FWD_ACC(c1, 1) RELATE_FWD_ACC(c0, 2) RELATE_OPS);
  else if constexpr (NW == 2 && FIRST) asm volatile(RELATE_FWD_SET(c0, 0) RELATE_FWD_SET(c1, 1) RELATE_OPS_FIRST);
  else if constexpr (NW == 2) asm volatile(RELATE_FWD_ACC(c0, 0) RELATE_FWD_ACC(c1, 1) RELATE_OPS);
  else asm volatile(RELATE_FWD_ACC(c0, 0) RELATE_OPS);
#undef RELATE_OPS_FIRST
#undef RELATE_OPS
}

// One site of the one-row recursion, state' = (state + rx) * (mismatch ? rho : 1), over the MPL words of
// the wave; c0 + c1 = sum of the new states.  The first round of row `rp` is already on its way into
// s[kBufA ..]; the first round of `rp_next` is requested after the last multiplication (with REC, the site
// record too: every scalar fetch of a site is then issued at one point, as far ahead of the next wait as the
// registers allow).
template <int MPL, int REC = 0>
__device__ __forceinline__ void sweep_row(double& c0, double& c1, double rx, double rho, uint64_t sk, const uint64_t* rp,
                                          const uint64_t* rp_next)
{
  using RR = RowRounds<MPL>;
  static_for<0, RR::NR>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    constexpr int w0 = RR::first(r), nw = RR::size(r);
    constexpr int NC = (nw + 3) / 4;  // groups of up to 4 words
    static_for<0, NC>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      const double rxl = rx;
      row_add<w0 + 4 * c, (nw - 4 * c < 4 ? nw - 4 * c : 4)>(rxl);
    });
    swait();
    static_for<0, NC>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      const double rhol = rho;
      const uint64_t skl = sk;
      row_mul<w0 + 4 * c, kBufA + 8 * c, (nw - 4 * c < 4 ? nw - 4 * c : 4)>(rhol, skl);
    });
    const uint64_t* rpl = rp;
    const uint64_t* rpn = rp_next;
    if constexpr (r + 1 < RR::NR) sload_round<RR::first(r + 1), RR::size(r + 1)>(rpl);
    else {
      sload_round<0, RR::size(0)>(rpn);
      // REC: the record of the site after the next travels with the row words (1: forward list order, 2: backward)
      if constexpr (REC == 1) prefetch_site<true>();
      if constexpr (REC == 2) prefetch_site<false>();
    }
    static_for<0, NC>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      row_acc<w0 + 4 * c, (nw - 4 * c < 4 ? nw - 4 * c : 4), r == 0 && c == 0>(c0, c1);
    });
  });
}

#define RELATE_BWD_FMA(i) \
  "s_andn2_b64 exec, %[skn], " RELATE_W(sn, i) "\n\tv_fma_f64 " RELATE_A(i) ", " RELATE_A(i) ", %[rho], %[c1]\n\t"
#define RELATE_BWD_MIS(c, i) \
  "s_andn2_b64 exec, %[skc], " RELATE_W(sc, i) "\n\tv_add_f64 %[" #c "], %[" #c "], " RELATE_A(i) "\n\t"

// The masked instructions of the two-row backward recursion on up to 4 words from state word W0 on.
// update: state = fma(state, rho, c1) where the lane mismatches the row handled one step earlier (masks s[SB ..], sk)
template <int W0, int SB, int NW>
__device__ __forceinline__ void row_fma(double rho, double c1, uint64_t sk)
{
#define RELATE_OPS : : [rho] "v"(rho), [c1] "v"(c1), [skn] "s"(sk), [r] "n"(kA0 + 2 * W0), [sn] "n"(SB) : "scc"
  if constexpr (NW == 4)
    asm volatile(RELATE_BWD_FMA(0) RELATE_BWD_FMA(1) RELATE_BWD_FMA(2) RELATE_BWD_FMA(3) "s_mov_b64 exec, -1" RELATE_OPS);
  else if constexpr (NW == 3) asm volatile(RELATE_BWD_FMA(0) RELATE_BWD_FMA(1) RELATE_BWD_FMA(2) "s_mov_b64 exec, -1" RELATE_OPS);
  else if constexpr (NW == 2) asm volatile(RELATE_BWD_FMA(0) RELATE_BWD_FMA(1) "s_mov_b64 exec, -1" RELATE_OPS);
  else asm volatile(RELATE_BWD_FMA(0) "s_mov_b64 exec, -1" RELATE_OPS);
#undef RELATE_OPS
}
// sums over the states that mismatch the current row (masks s[SB ..], sk); the caller zeroes m0, m1 per site
template <int W0, int SB, int NW>
__device__ __forceinline__ void row_acc_mis(double& m0, double& m1, uint64_t sk)
{
#define RELATE_OPS : [cm0] "+v"(m0), [cm1] "+v"(m1) : [skc] "s"(sk), [r] "n"(kA0 + 2 * W0), [sc] "n"(SB) : "scc"
  if constexpr (NW == 4)
    asm volatile(RELATE_BWD_MIS(cm0, 0) RELATE_BWD_MIS(cm1, 1) RELATE_BWD_MIS(cm0, 2) RELATE_BWD_MIS(cm1, 3) "s_mov_b64 exec, -1" RELATE_OPS);
  else if constexpr (NW == 3)
    asm volatile(RELATE_BWD_MIS(cm0, 0) RELATE_BWD_MIS(cm1, 1) RELATE_BWD_MIS(cm0, 2) "s_mov_b64 exec, -1" RELATE_OPS);
  else if constexpr (NW == 2) asm volatile(RELATE_BWD_MIS(cm0, 0) RELATE_BWD_MIS(cm1, 1) "s_mov_b64 exec, -1" RELATE_OPS);
  else asm volatile(RELATE_BWD_MIS(cm0, 0) "s_mov_b64 exec, -1" RELATE_OPS);
#undef RELATE_OPS
}

// The two-row recursion takes the masks of two rows per word, so a round holds half as many words of each (8, or 10
// where the compiler leaves 40 registers to the row words): those of the row handled one step earlier in
// s[kBufA ..], those of the current row in s[kBufC ..].
// words [W0, W0 + NW) of the row at p -> s[SB ..]: eight as one load, the ninth and tenth as a load of their own
template <int SB, int W0, int NW>
__device__ __forceinline__ void sload_row2(const uint64_t* p)
{
  static_assert(NW >= 1 && NW <= kBufWords2 && NW <= 10, "one round of one row");
  sload_words<SB, 8 * W0, (NW < 8 ? NW : 8)>(p);
  if constexpr (NW > 8) asm volatile("s_load_dwordx4 s[%1:%1+3], %0, %2" ::"s"(p), "n"(SB + 16), "n"(8 * (W0 + 8)) : "memory");
}
template <int MPL>
struct RowRounds2 {
  static constexpr int NR = (MPL + kBufWords2 - 1) / kBufWords2;
  static constexpr int size(int r) { return MPL / NR + (r < MPL % NR ? 1 : 0); }
  static constexpr int first(int r)
  {
    int o = 0;
    for (int i = 0; i < r; i++) o += size(i);
    return o;
  }
};
// One site of the two-row backward recursion (src/fast_painting.cpp:488-517): t = beta + aa for every state,
// fma(t, rho, c1) where the state mismatches row `rp_n`, then the sums of the new states over all of them
// (a0 + a1) and over those that mismatch row `rp_c` (m0 + m1).  Rounds as in sweep_row: additions -- wait --
// the two masked kinds -- fetch of the next round -- plain sums.  The first round of (rp_n, rp_c) is on its
// way already; the first round of the next site, (rp_c, rp_p), is requested in the last round.
template <int MPL>
__device__ __forceinline__ void sweep_two_rows(double& a0, double& a1, double& m0, double& m1, double aa, double rho, double c1,
                                               uint64_t skn, uint64_t skc, const uint64_t* rp_n, const uint64_t* rp_c,
                                               const uint64_t* rp_p)
{
  using RR = RowRounds2<MPL>;
  static_for<0, RR::NR>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    constexpr int w0 = RR::first(r), nw = RR::size(r);
    constexpr int NC = (nw + 3) / 4;
    static_for<0, NC>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      const double aal = aa;
      row_add<w0 + 4 * c, (nw - 4 * c < 4 ? nw - 4 * c : 4)>(aal);
    });
    swait();
    static_for<0, NC>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      const double rhol = rho, c1l = c1;
      const uint64_t skl = skn;
      row_fma<w0 + 4 * c, kBufA + 8 * c, (nw - 4 * c < 4 ? nw - 4 * c : 4)>(rhol, c1l, skl);
    });
    static_for<0, NC>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      const uint64_t skl = skc;
      row_acc_mis<w0 + 4 * c, kBufC + 8 * c, (nw - 4 * c < 4 ? nw - 4 * c : 4)>(m0, m1, skl);
    });
    const uint64_t* pn = rp_n;
    const uint64_t* pc = rp_c;
    const uint64_t* pq = rp_p;
    if constexpr (r + 1 < RR::NR) {
      sload_row2<kBufA, RR::first(r + 1), RR::size(r + 1)>(pn);
      sload_row2<kBufC, RR::first(r + 1), RR::size(r + 1)>(pc);
    } else {
      sload_row2<kBufA, 0, RR::size(0)>(pc);
      sload_row2<kBufC, 0, RR::size(0)>(pq);
    }
    static_for<0, NC>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      row_acc<w0 + 4 * c, (nw - 4 * c < 4 ? nw - 4 * c : 4), r == 0 && c == 0>(a0, a1);
    });
  });
}

// ---- reductions ------------------------------------------------------------------------------------
#define RELATE_DPP_STEP(ctrl)                                                            \
  "s_nop 1\n\t"                                                                          \
  "v_mov_b32_dpp v[%[t]], v[%[x]] " ctrl " row_mask:0xf bank_mask:0xf\n\t"               \
  "v_mov_b32_dpp v[%[t]+1], v[%[x]+1] " ctrl " row_mask:0xf bank_mask:0xf\n\t"           \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[t]:%[t]+1]\n\t"

// Sum over the 64 lanes of the value left in v[kX:kX+1], the same bits in every lane of the result: four
// exchange steps inside each 16-lane row, then the row sums travel down the wave (row_bcast 15: rows 1 and 3
// take in rows 0 and 2; row_bcast 31: row 3 takes in row 1), i.e. (r3 + r2) + (r1 + r0), and lane 63 is read out.
#define RELATE_WAVE_SUM_TAIL                                                                   \
  RELATE_DPP_STEP("quad_perm:[1,0,3,2]")                                                       \
  RELATE_DPP_STEP("quad_perm:[2,3,0,1]")                                                       \
  RELATE_DPP_STEP("row_half_mirror")                                                           \
  RELATE_DPP_STEP("row_mirror")                                                                \
  "s_nop 1\n\t"                                                                                \
  "v_mov_b32_dpp v[%[t]], v[%[x]] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                 \
  "v_mov_b32_dpp v[%[t]+1], v[%[x]+1] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"             \
  "s_mov_b32 exec_lo, 0xffff0000\n\t"                                                         \
  "s_mov_b32 exec_hi, 0xffff0000\n\t"                                                         \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[t]:%[t]+1]\n\t"                               \
  "s_mov_b64 exec, -1\n\t"                                                                     \
  "s_nop 0\n\t"                                                                                \
  "v_mov_b32_dpp v[%[t]], v[%[x]] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                 \
  "v_mov_b32_dpp v[%[t]+1], v[%[x]+1] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"             \
  "s_mov_b32 exec_lo, 0\n\t"                                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[t]:%[t]+1]\n\t"                               \
  "s_mov_b64 exec, -1\n\t"                                                                     \
  "s_nop 0\n\t"                                                                                \
  "v_readlane_b32 vcc_lo, v[%[x]], 63\n\t"                                                     \
  "v_readlane_b32 vcc_hi, v[%[x]+1], 63\n\t"                                                   \
  "s_nop 1\n\t"                                                                                \
  "v_mov_b64 %[out], vcc\n\t"                                                                  \
  "s_mov_b32 %[hi], vcc_hi"

// wave sum of x; `hi` = its high word as a scalar (the lane-63 read-out passes through VCC: the row-word
// registers may be receiving the next row by then)
__device__ __forceinline__ double wave_sum_regs(uint32_t& hi, double x)
{
  double out;
  asm volatile("v_mov_b64 v[%[x]:%[x]+1], %[in]\n\t" RELATE_WAVE_SUM_TAIL
               : [out] "=v"(out), [hi] "=s"(hi)
               : [in] "v"(x), [x] "n"(kX), [t] "n"(kT)
               : "vcc");
  return out;
}
// wave sum of a + b
__device__ __forceinline__ double wave_sum_regs(uint32_t& hi, double a, double b)
{
  double out;
  asm volatile("v_add_f64 v[%[x]:%[x]+1], %[a], %[b]\n\t" RELATE_WAVE_SUM_TAIL
               : [out] "=v"(out), [hi] "=s"(hi)
               : [a] "v"(a), [b] "v"(b), [x] "n"(kX), [t] "n"(kT)
               : "vcc");
  return out;
}
// wave sum of wa * a + wb * b with the weights in scalar registers
struct ScalarWeight {
  double w;
};
__device__ __forceinline__ double wave_sum_regs(uint32_t& hi, double a, ScalarWeight wa, double b, ScalarWeight wb)
{
  double out;
  asm volatile("v_mul_f64 v[%[x]:%[x]+1], %[b], %[wb]\n\t"
               "v_fma_f64 v[%[x]:%[x]+1], %[a], %[wa], v[%[x]:%[x]+1]\n\t" RELATE_WAVE_SUM_TAIL
               : [out] "=v"(out), [hi] "=s"(hi)
               : [a] "v"(a), [wa] "s"(wa.w), [b] "v"(b), [wb] "s"(wb.w), [x] "n"(kX), [t] "n"(kT)
               : "vcc");
  return out;
}
// wave sum of wa * a + wb * b
__device__ __forceinline__ double wave_sum_regs(uint32_t& hi, double a, double wa, double b, double wb)
{
  double out;
  asm volatile("v_mul_f64 v[%[x]:%[x]+1], %[b], %[wb]\n\t"
               "v_fma_f64 v[%[x]:%[x]+1], %[a], %[wa], v[%[x]:%[x]+1]\n\t" RELATE_WAVE_SUM_TAIL
               : [out] "=v"(out), [hi] "=s"(hi)
               : [a] "v"(a), [wa] "v"(wa), [b] "v"(b), [wb] "v"(wb), [x] "n"(kX), [t] "n"(kT)
               : "vcc");
  return out;
}

// Sum over the G waves of the workgroup; `slot` alternates between two LDS rows so that one barrier
// per call is enough (a row is rewritten two calls later, after every wave has passed the barrier of
// the call in between).
template <int G, class... Parts>
__device__ __forceinline__ double block_sum_hi(double (*red)[G], int& slot, uint32_t& hi, Parts... parts)
{
  double x = wave_sum_regs(hi, parts...);
  if constexpr (G == 1) {
    return x;
  } else {
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 0) red[slot][wv] = x;
    __syncthreads();
    double s = red[slot][0];
#pragma unroll
    for (int g = 1; g < G; g++) s += red[slot][g];
    slot ^= 1;
    hi = sgpr32((uint32_t)__double2hiint(s));
    return s;
  }
}
// Sum of a + b over the G > 1 waves of the workgroup, for the one-row loops.  After the last exchange step of the
// wave sum the 16 lanes of row 3 hold it; one of them writes it to the wave's LDS entry (no read-out into
// every lane), and after the barrier every wave adds the G entries in the same order.  Two sets of entries
// alternate with the parity of the loop counter `step`, so that one barrier per call is enough (the caller
// puts a barrier between the last use of the entries before its loop and the loop); which set is written is a
// matter of which lane writes -- lane 63 holds the address of the wave's entry in set 0, lane 62 in set 1
// (`wr_addr`, see exchange_addresses) -- and the reads of the two sets are two copies of the code with immediate
// offsets behind a scalar branch: no vector instruction computes an address.  No scalar load may be in flight
// after the wait before the barrier: the counted waits of the reads rely on LDS order.  The entries pass through
// v[kX], v[kT] and v[kHi], which the wave sum has finished with.
#define RELATE_XCHG_FIRST "v_add_f64 v[%[x]:%[x]+1], %[a], %[b]\n\t"
// (the two-row backward recursion sums wa * a + wb * b, the weights in scalar registers)
#define RELATE_XCHG_FIRST_W \
  "v_mul_f64 v[%[x]:%[x]+1], %[b], %[kb]\n\t"  \
  "v_fma_f64 v[%[x]:%[x]+1], %[a], %[ka], v[%[x]:%[x]+1]\n\t"
#define RELATE_XCHG_HEAD                                                                       \
  RELATE_DPP_STEP("quad_perm:[1,0,3,2]")                                                       \
  RELATE_DPP_STEP("quad_perm:[2,3,0,1]")                                                       \
  RELATE_DPP_STEP("row_half_mirror")                                                           \
  RELATE_DPP_STEP("row_mirror")                                                                \
  "s_nop 1\n\t"                                                                                \
  "v_mov_b32_dpp v[%[t]], v[%[x]] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                 \
  "v_mov_b32_dpp v[%[t]+1], v[%[x]+1] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"             \
  "s_mov_b32 exec_lo, 0xffff0000\n\t"                                                         \
  "s_mov_b32 exec_hi, 0xffff0000\n\t"                                                         \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[t]:%[t]+1]\n\t"                               \
  "s_mov_b64 exec, -1\n\t"                                                                     \
  "s_nop 0\n\t"                                                                                \
  "v_mov_b32_dpp v[%[t]], v[%[x]] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                 \
  "v_mov_b32_dpp v[%[t]+1], v[%[x]+1] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"             \
  "s_mov_b32 exec_lo, 0\n\t"                                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[t]:%[t]+1]\n\t"                               \
  "s_and_b32 vcc_lo, %[step], 1\n\t"                                                          \
  "s_lshr_b32 exec_hi, 0x80000000, vcc_lo\n\t"                                                \
  "ds_write_b64 %[wa], v[%[x]:%[x]+1]\n\t"                                                     \
  "s_mov_b64 exec, -1\n\t"                                                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                   \
  "s_barrier\n\t"                                                                              \
  "s_bitcmp0_b32 %[step], 0\n\t"                                                               \
  "s_cbranch_scc0 .Lxs1%=\n\t"
// the entries of one set: reads into v[kX], v[kT], v[kHi], the last one into the result register; ((e0 + e1) + e2) + e3
#define RELATE_XCHG_SET2(o)                                                   \
  "ds_read_b64 v[%[x]:%[x]+1], %[rb] offset:" #o "\n\t"                        \
  "ds_read_b64 %[s], %[rb] offset:" #o "+8\n\t"                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                   \
  "v_add_f64 %[s], v[%[x]:%[x]+1], %[s]\n\t"
#define RELATE_XCHG_SET4(o)                                                   \
  "ds_read_b64 v[%[x]:%[x]+1], %[rb] offset:" #o "\n\t"                        \
  "ds_read_b64 v[%[t]:%[t]+1], %[rb] offset:" #o "+8\n\t"                      \
  "ds_read_b64 v[%[h]:%[h]+1], %[rb] offset:" #o "+16\n\t"                     \
  "ds_read_b64 %[s], %[rb] offset:" #o "+24\n\t"                               \
  "s_waitcnt lgkmcnt(2)\n\t"                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[t]:%[t]+1]\n\t"               \
  "s_waitcnt lgkmcnt(1)\n\t"                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[h]:%[h]+1]\n\t"               \
  "s_waitcnt lgkmcnt(0)\n\t"                                                   \
  "v_add_f64 %[s], v[%[x]:%[x]+1], %[s]\n\t"
// eight entries: the first four as above into v[kX], then four more through v[kT], v[kHi] and the result register
#define RELATE_XCHG_SET8(o)                                                   \
  "ds_read_b64 v[%[x]:%[x]+1], %[rb] offset:" #o "\n\t"                        \
  "ds_read_b64 v[%[t]:%[t]+1], %[rb] offset:" #o "+8\n\t"                      \
  "ds_read_b64 v[%[h]:%[h]+1], %[rb] offset:" #o "+16\n\t"                     \
  "ds_read_b64 %[s], %[rb] offset:" #o "+24\n\t"                               \
  "s_waitcnt lgkmcnt(2)\n\t"                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[t]:%[t]+1]\n\t"               \
  "ds_read_b64 v[%[t]:%[t]+1], %[rb] offset:" #o "+32\n\t"                     \
  "s_waitcnt lgkmcnt(2)\n\t"                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[h]:%[h]+1]\n\t"               \
  "ds_read_b64 v[%[h]:%[h]+1], %[rb] offset:" #o "+40\n\t"                     \
  "s_waitcnt lgkmcnt(2)\n\t"                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], %[s]\n\t"                         \
  "ds_read_b64 %[s], %[rb] offset:" #o "+48\n\t"                               \
  "s_waitcnt lgkmcnt(2)\n\t"                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[t]:%[t]+1]\n\t"               \
  "ds_read_b64 v[%[t]:%[t]+1], %[rb] offset:" #o "+56\n\t"                     \
  "s_waitcnt lgkmcnt(2)\n\t"                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[h]:%[h]+1]\n\t"               \
  "s_waitcnt lgkmcnt(1)\n\t"                                                   \
  "v_add_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], %[s]\n\t"                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                   \
  "v_add_f64 %[s], v[%[x]:%[x]+1], v[%[t]:%[t]+1]\n\t"
#define RELATE_XCHG_ASM(SET, o1)                                                                                      \
  asm volatile(RELATE_XCHG_FIRST RELATE_XCHG_HEAD SET(0) "s_branch .Lxsd%=\n.Lxs1%=:\n\t" SET(o1) ".Lxsd%=:"           \
               : [s] "=&v"(s)                                                                                         \
               : [a] "v"(a), [b] "v"(b), [wa] "v"(wr_addr), [rb] "v"(rd_base), [step] "s"(step), [x] "n"(kX), [t] "n"(kT), \
                 [h] "n"(kHi)                                                                                         \
               : "memory", "scc", "vcc")
#define RELATE_XCHG_ASM_W(SET, o1)                                                                                    \
  asm volatile(RELATE_XCHG_FIRST_W RELATE_XCHG_HEAD SET(0) "s_branch .Lxsd%=\n.Lxs1%=:\n\t" SET(o1) ".Lxsd%=:"         \
               : [s] "=&v"(s)                                                                                         \
               : [a] "v"(a), [b] "v"(b), [ka] "s"(wa.w), [kb] "s"(wb.w), [wa] "v"(wr_addr), [rb] "v"(rd_base),        \
                 [step] "s"(step), [x] "n"(kX), [t] "n"(kT), [h] "n"(kHi)                                              \
               : "memory", "scc", "vcc")
template <int G>
__device__ __forceinline__ double exchange_sum(uint32_t wr_addr, uint32_t rd_base, int step, double a, double b)
{
  static_assert(G == 2 || G == 4 || G == 8, "waves per target");
  double s;
  if constexpr (G == 2) RELATE_XCHG_ASM(RELATE_XCHG_SET2, 16);
  else if constexpr (G == 4) RELATE_XCHG_ASM(RELATE_XCHG_SET4, 32);
  else RELATE_XCHG_ASM(RELATE_XCHG_SET8, 64);
  return s;
}
// the same for wa * a + wb * b
template <int G>
__device__ __forceinline__ double exchange_sum(uint32_t wr_addr, uint32_t rd_base, int step, double a, ScalarWeight wa, double b, ScalarWeight wb)
{
  static_assert(G == 2 || G == 4 || G == 8, "waves per target");
  double s;
  if constexpr (G == 2) RELATE_XCHG_ASM_W(RELATE_XCHG_SET2, 16);
  else if constexpr (G == 4) RELATE_XCHG_ASM_W(RELATE_XCHG_SET4, 32);
  else RELATE_XCHG_ASM_W(RELATE_XCHG_SET8, 64);
  return s;
}
// LDS addresses for exchange_sum: `red` = the 2 x G entries; lane 62 writes set 1, every other lane's address is
// that of set 0 (only lane 63's is used).
template <int G>
__device__ __forceinline__ void exchange_addresses(double (*red)[G], uint32_t& wr_addr, uint32_t& rd_base)
{
  const int wv = threadIdx.x >> 6;
  rd_base = (uint32_t)(uintptr_t)&red[0][0];
  wr_addr = (uint32_t)(uintptr_t)&red[lane_id() == 62 ? 1 : 0][wv];
  asm volatile("" : "+v"(wr_addr), "+v"(rd_base));  // stay in vector registers across the loop
}

template <int G, class... Parts>
__device__ __forceinline__ double block_sum_regs(double (*red)[G], int& slot, Parts... parts)
{
  uint32_t hi;
  return block_sum_hi<G>(red, slot, hi, parts...);
}

// The instruction arbiter of a SIMD serves the oldest wave first, so of the four waves that share a SIMD
// the youngest would lag and run the end of its pass alone, at the issue rate of a single wave.  Every
// wave therefore rotates its own priority with its progress (rank = its slot on the SIMD, i.e. its age
// rank): all four see the same sequence of priorities and advance at the same average rate.
// The backward pass has 4/3 of the work of the forward pass per site, so its waves (`bias` 1) move up
// one level three quarters of the time and the two kinds end closer together.
__device__ __forceinline__ void rotate_priority(int rank, int j, int bias)
{
  int level = (rank + (j >> 5)) & 3;
  if (bias && ((j >> 7) & 3) && level < 3) level++;
  switch (level) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// ---- kernel ----------------------------------------------------------------------------------------
// Kernel arguments as one struct, so that the rarely used ones can be fetched from the kernel-argument
// segment where they are needed instead of occupying scalar registers across the sweeps.
struct KArgs {
  const uint64_t* rows;
  const int32_t* dk_all;
  const SiteRec* rec_all;
  const int64_t* off;
  const int32_t* nd_all;
  const int32_t* bbj_all;
  const int32_t* bej_all;
  const int32_t* order;
  const int32_t* wb;
  uint8_t* out;
  uint64_t* trace;
  Scal p;
};
typedef const __attribute__((address_space(4))) KArgs* KArgsPtr;
// The argument segment again, as a pointer the optimiser knows nothing about: loads through it stay
// where they are written (inside the stone / rescale paths) and are not hoisted above the loops.
__device__ __forceinline__ KArgsPtr rare_args()
{
  KArgsPtr q = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return q;
}

// S < lo || S > hi with the thresholds fetched where they are needed (rare: see surely_in_range)
__device__ __forceinline__ bool out_of_range_exact(double S)
{
  KArgsPtr q = rare_args();
  return __builtin_amdgcn_ballot_w64(S < q->p.c.lo || S > q->p.c.hi) != 0;  // S is the same in every lane
}

template <int MPL, int G>
__global__ void RELATE_REGS_ATTR __launch_bounds__(64 * G) paint_kernel_r(const KArgs ka)
{
  static_assert(MPL >= 4 && MPL <= kMaxMPL, "states per lane");
  // Dynamic LDS (sized by the launcher): with a static allocation the compiler derives an occupancy
  // bound from it and then ignores amdgpu_num_sgpr.  Two rows of G partial sums, G window
  // counters, then the staging buffer of one stone payload (G * MPL * 64 floats).
  extern __shared__ double lds_dyn[];
  double (*red)[G] = reinterpret_cast<double (*)[G]>(lds_dyn);
  // The window whose stone comes next is looked at only when a stone is written: it lives in LDS (one slot
  // per wave, read and written by that wave alone) instead of a scalar register that the sweeps would carry.
  volatile int* wslot = reinterpret_cast<volatile int*>(lds_dyn + 2 * G) + 2 * (threadIdx.x >> 6);
  auto next_window = [&]() __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*wslot); };
  float* stage = reinterpret_cast<float*>(lds_dyn + 3 * G);
  claim_registers<MPL>();
  int slot = 0;
  // The two passes of a target are independent until their stones meet in the record, so they run as
  // two workgroups: twice as many waves to interleave on a SIMD, with nothing to exchange.  Which of
  // the pair 2i, 2i+1 takes which pass follows the parity of the set bits of the block index, so that
  // every power-of-two stride of the dispatcher (XCDs, CUs, SIMDs) sees both kinds.
  const int pass = __builtin_popcount(blockIdx.x) & 1;
  // slot of this wave on its SIMD (HW_ID.wave_id); the waves of a multi-wave workgroup meet at a barrier
  // every site and must share one schedule, so they take it from the block index instead
  const int rank = G == 1 ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | 4) & 3) : (int)((blockIdx.x >> 1) & 3);
  const int t = __builtin_amdgcn_readfirstlane(ka.order[blockIdx.x >> 1]);
  const int k = ka.p.k_begin + t;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wbase = wave * MPL;
  const int stride = ka.p.row_stride;
  const int nd = __builtin_amdgcn_readfirstlane(ka.nd_all[t]);
  const long long o = ka.off[t];
  const int32_t* __restrict__ dk = ka.dk_all + o;
  const SiteRec* __restrict__ rec = ka.rec_all + o;
  const int kw = k >> 6, kb = k & 63;
  const bool first_thread = threadIdx.x == 0;
  const uint64_t* rows_w = ka.rows + wbase;  // the words of this wave start here in every row
  // diagnostics: the start time goes out now, so that no loop carries it (finish() writes the rest)
  if (ka.trace && first_thread) ka.trace[(size_t)4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();

  // Lanes without a state: the target's own (its bit, or 0 if its word belongs to another wave) and the
  // padding lanes from the first incomplete word of this wave on; into their fixed registers.
  auto stateless = [&]() __attribute__((always_inline)) {
    // (through the opaque pointer: read from `ka`, the compiler may fetch the word at the latch of the site loop
    // that follows, with a wait for it at every site)
    const int N = rare_args()->p.N;
    const int full_words = N >> 6, rem = N & 63;
    const bool self_here = kw >= wbase && kw < wbase + MPL;
    const int w_ph = full_words > wbase ? full_words : wbase;
    set_stateless(sgpr64(self_here ? (1ull << kb) : 0ull), sgpr32(self_here ? 2u * (uint32_t)(kw - wbase) : 0u),
                  sgpr64((w_ph == full_words && rem) ? ~((1ull << rem) - 1ull) : ~0ull),
                  sgpr32(w_ph - wbase < MPL ? 2u * (uint32_t)(w_ph - wbase) : 2u * MPL), 2u * MPL);
  };

  // bit of the target itself at the two forced sites; it is '1' at every other site of its list
  const uint64_t sk_first = sgpr64(((ka.rows[kw] >> kb) & 1ull) ? ~0ull : 0ull);
  const uint64_t sk_last = sgpr64(((ka.rows[(size_t)(ka.p.L - 1) * stride + kw] >> kb) & 1ull) ? ~0ull : 0ull);

  const uint32_t lds_mine = (uint32_t)(uintptr_t)(stage + wbase * 64 + lane);

  // Stone payload (src/fast_painting.cpp:368-388,:573-592): every lane drops its states as floats into
  // LDS, then the workgroup streams the N floats out with coalesced stores; the target's own entry is
  // an exact zero already (:312,:507).
  auto store_states = [&](float* __restrict__ dst, int N) __attribute__((always_inline)) {
    const uint32_t a = lds_mine;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) { state_to_lds<decltype(ic)::value>(a); });
    swait();
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 64 * G) dst[n] = stage[n];
    __syncthreads();
  };

  // Diagnostics at the end of a pass.  Each pass ends with its own call and returns: code shared after the
  // three branches would let the compiler move the set-up loads of a later branch into the loop of an earlier one.
  // What it writes is fetched again here (the list length) or computed again (the pass): a value kept for it
  // through the loops costs them a register, and a spilled one a vector instruction per site to bring it back.
  auto finish = [&]() __attribute__((always_inline)) {
    KArgsPtr q = rare_args();
    if (q->trace && threadIdx.x == 0) {
      uint64_t* tr = q->trace + (size_t)4 * blockIdx.x;
      const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      tr[0] = ((uint64_t)xcc << 32) | hw;
      tr[2] = __builtin_amdgcn_s_memrealtime();
      tr[3] = (uint64_t)(__builtin_popcount(blockIdx.x) & 1) | ((uint64_t)q->nd_all[q->order[blockIdx.x >> 1]] << 8);
    }
  };

  // ---------------- forward (src/fast_painting.cpp:214-408) ----------------
  if (pass == 0) {
    double acc = 0.0;
    const double prior_mis = ka.p.c.prior_mis, prior_match = ka.p.c.prior_match;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int wi = wbase + i;
      const uint64_t w = rows_w[i];
      const bool v = __builtin_amdgcn_inverse_ballot_w64(valid_mask(wi, ka.p.N, k));
      const bool m = __builtin_amdgcn_inverse_ballot_w64(~w & sk_first);
      const double a = m ? prior_mis : prior_match;
      state_set<i>(v ? a : 0.0);
      acc += v ? a : 0.0;
    });
    double S = block_sum_regs<G>(red, slot, acc);
    double ls = 0.0;
    *wslot = 0;
    // The loop counts the sites still to come (left = nd - 1 - j), so that nd itself is not needed in a scalar
    // register at every site; the next alpha stone is due when `left` reaches stone_left (window 0 starts at site 0).
    int stone_left = nd - 1;
    auto emit = [&](int left) __attribute__((always_inline)) {
      while (stone_left == left) {
        KArgsPtr q = rare_args();
        const int N = q->p.N, W = q->p.W;
        int wcur = next_window();
        uint8_t* rec = q->out + ((size_t)wcur * q->p.K + t) * (56 + (size_t)8 * N);
        if (first_thread) {
          put_u32(rec, (uint32_t)q->wb[wcur]);
          put_u32(rec + 4, (uint32_t)(q->wb[wcur + 1] - 1));
          put_block_header(rec + 8, N, q->dk_all[q->off[t] + (q->nd_all[t] - 1 - left)], (float)ls);
        }
        store_states(reinterpret_cast<float*>(rec + 32), N);
        wcur++;
        *wslot = wcur;
        stone_left = wcur < W ? __builtin_amdgcn_readfirstlane(q->nd_all[t] - 1 - q->bbj_all[(long long)t * W + wcur]) : -1;
      }
    };
    emit(nd - 1);
    stateless();
    set_rows_base(rows_w);
    set_edge_mask(sk_last);
    uint32_t stride_bytes = 8u * (uint32_t)stride;
    asm volatile("" : "+s"(stride_bytes));  // not to be fetched again inside the loop: a scalar load there brings a wait with it
    set_touch_lane(4u * (uint32_t)lane);
    double rho = ka.p.c.rho_theta;
    asm volatile("" : "+v"(rho));  // stays in a vector register: re-reading it from the argument segment would add a wait to every site
    double coef_j = sgpr_f64(rec[0].coef), nx_j = sgpr_f64(rec[0].nx);  // terms of the gap before site j
    if constexpr (G > 1) set_site_terms(coef_j, nx_j);
    set_site_pointer(rec + 1);
    // nd >= 2: sites 0 and L-1 are always listed
    const uint64_t* rp = row_pointer(__builtin_amdgcn_readfirstlane(dk[1]), stride_bytes);
    const uint64_t* rp_next = row_pointer(__builtin_amdgcn_readfirstlane(dk[2 < nd ? 2 : nd - 1]), stride_bytes);
    sload_round<0, RowRounds<MPL>::size(0)>(rp);
    uint32_t wr_addr = 0, rd_base = 0;
    if constexpr (G > 1) {
      exchange_addresses<G>(red, wr_addr, rd_base);
      __syncthreads();  // every wave has read the entries of the sum above
    }
    for (int left = nd - 2; left >= 0; left--) {
      if ((left & 31) == 0) rotate_priority(rank, left, 0);
      // scalars of site j+1 travel during this sweep: the list entry two sites on and the terms of the gap before
      // it.  One wave per target: fetched before the sweep, copied into compiler registers after it.  Several: fetched
      // with the next site's row words, after the last use of this site's (the wait before the barrier is then the
      // only one of the site that finds a scalar fetch in flight), and used where they arrive.
      double rx, c0, c1;
      uint32_t S_high;
      const uint64_t* rp_next2;
      if constexpr (G == 1) {
        rx = S * coef_j;
        write_stateless<MPL, true>(S, coef_j);
        prefetch_site<true>();
        const uint64_t skm = edge_mask(left, 0);
        ls += nx_j;
        sweep_row<MPL>(c0, c1, rx, rho, skm, rp, rp_next);
        S = block_sum_hi<G>(red, slot, S_high, c0, c1);
        int s_next2;
        fetched_site<true>(s_next2, coef_j, nx_j);  // arrived: the sweep has waited since the fetch
        rp_next2 = row_pointer(s_next2, stride_bytes);
      } else {
        rx = times_site_coef(S);
        write_stateless_site<MPL>(S);
        const uint64_t skm = edge_mask(left, 0);
        add_site_nx(ls);
        sweep_row<MPL, 1>(c0, c1, rx, rho, skm, rp, rp_next);
        S = exchange_sum<G>(wr_addr, rd_base, left, c0, c1);
        S_high = sgpr32((uint32_t)__double2hiint(S));
        rp_next2 = fetched_row_pointer<true>(stride_bytes);
      }
      if constexpr (kTouchRows) touch_row<MPL>(rp_next2);
      if (!surely_in_range(S_high) && out_of_range_exact(S)) {
        const double inv = 1.0 / S;
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
          const double f = inv;
          state_scale<decltype(ic)::value>(f);
        });
        ls += log(S);
        S = 1.0;
      }
      emit(left);
      rp = rp_next;
      rp_next = rp_next2;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    finish();
    return;
  }

  // ---------------- backward, one emission probability (src/fast_painting.cpp:410-599) ----------------
  // The reference keeps beta and weights it with the emission of the current site when it sums
  // (:509-517), which takes the masks of two rows per step.  When the painter and the data hold the same
  // theta (FastPainting(Data&), src/fast_painting.hpp:32-45) the emission-weighted state g = beta * e obeys
  // the forward recursion, g' = (g + rx) * (mismatch' ? rho : 1) with B = sum g, on the masks of one row;
  // beta itself is needed at the stones only, as g / e.
  else if (ka.p.c.same_theta) {
    const double theta_d = ka.p.c.theta_d, ntheta_d = ka.p.c.ntheta_d;
    double ls = ka.p.c.log_Nm1 - (double)nd * ka.p.c.log_ntheta;
    const uint64_t* row_last = rows_w + (size_t)(ka.p.L - 1) * stride;
    double acc = 0.0;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int wi = wbase + i;
      const uint64_t w = row_last[i];
      const bool v = __builtin_amdgcn_inverse_ballot_w64(valid_mask(wi, ka.p.N, k));
      const bool m = __builtin_amdgcn_inverse_ballot_w64(~w & sk_last);
      const double g = v ? (m ? theta_d : ntheta_d) : 0.0;
      state_set<i>(g);
      acc += g;
    });
    double B = block_sum_regs<G>(red, slot, acc);
    *wslot = ka.p.W - 1;
    int stone_j = nd - 1;  // the last window ends at the last SNP
    // stones taken at the last SNP hold beta = 1 for every state, the target included (:427-432,:447-462)
    while (stone_j == nd - 1) {
      KArgsPtr q = rare_args();
      const int N = q->p.N;
      int wcur = next_window();
      uint8_t* rec = q->out + ((size_t)wcur * q->p.K + t) * (56 + (size_t)8 * N);
      if (first_thread) put_block_header(rec + 32 + (size_t)4 * N, N, q->p.L - 1, (float)ls);
      float* dst = reinterpret_cast<float*>(rec + 56 + (size_t)4 * N);
      for (int n = threadIdx.x; n < N; n += 64 * G) dst[n] = 1.0f;
      wcur--;
      *wslot = wcur;
      stone_j = wcur >= 0 ? __builtin_amdgcn_readfirstlane(q->bej_all[(long long)t * q->p.W + wcur]) : -1;
    }
    stateless();
    set_rows_base(rows_w);
    set_edge_mask(sk_first);
    uint32_t stride_bytes = 8u * (uint32_t)stride;
    asm volatile("" : "+s"(stride_bytes));  // not to be fetched again inside the loop: a scalar load there brings a wait with it
    set_touch_lane(4u * (uint32_t)lane);
    double rho = ka.p.c.rho_theta;
    asm volatile("" : "+v"(rho));
    double coef_j = sgpr_f64(rec[nd - 1].coef), nx_j = sgpr_f64(rec[nd - 1].nx);  // terms used by the step to site j: index j + 1
    if constexpr (G > 1) set_site_terms(coef_j, nx_j);
    set_site_pointer(rec + (nd - 2));
    const uint64_t* rp = row_pointer(__builtin_amdgcn_readfirstlane(dk[nd - 2]), stride_bytes);
    const uint64_t* rp_next = row_pointer(__builtin_amdgcn_readfirstlane(dk[nd > 2 ? nd - 3 : 0]), stride_bytes);
    sload_round<0, RowRounds<MPL>::size(0)>(rp);
    uint32_t wr_addr = 0, rd_base = 0;
    if constexpr (G > 1) {
      exchange_addresses<G>(red, wr_addr, rd_base);
      __syncthreads();  // every wave has read the entries of the sum above
    }
    for (int j = nd - 2; j >= 0; j--) {
      if ((j & 31) == 0) rotate_priority(rank, j, 0);
      // scalars of the step to site j-1 travel during this sweep (see the forward loop)
      double rx, c0, c1;
      uint32_t B_high;
      const uint64_t* rp_next2;
      const uint64_t skm = edge_mask(j, 0);
      if constexpr (G == 1) {
        rx = B * coef_j;
        write_stateless<MPL, true>(B, coef_j);
        prefetch_site<false>();
        ls += nx_j;
        sweep_row<MPL>(c0, c1, rx, rho, skm, rp, rp_next);
        B = block_sum_hi<G>(red, slot, B_high, c0, c1);
        int s_prev2;
        fetched_site<false>(s_prev2, coef_j, nx_j);
        rp_next2 = row_pointer(s_prev2, stride_bytes);
      } else {
        rx = times_site_coef(B);
        write_stateless_site<MPL>(B);
        add_site_nx(ls);
        sweep_row<MPL, 2>(c0, c1, rx, rho, skm, rp, rp_next);
        B = exchange_sum<G>(wr_addr, rd_base, j, c0, c1);
        B_high = sgpr32((uint32_t)__double2hiint(B));
        rp_next2 = fetched_row_pointer<false>(stride_bytes);
      }
      if constexpr (kTouchRows) touch_row<MPL>(rp_next2);
      if (!surely_in_range(B_high) && out_of_range_exact(B)) {
        const double inv = 1.0 / B;
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
          const double f = inv;
          state_scale<decltype(ic)::value>(f);
        });
        ls += (double)fast_log_dev((float)B);  // float approximation on this side (:562)
        B = 1.0;
      }
      while (stone_j == j) {
        KArgsPtr q = rare_args();
        const int N = q->p.N;
        int wcur = next_window();
        uint8_t* rec = q->out + ((size_t)wcur * q->p.K + t) * (56 + (size_t)8 * N);
        if (first_thread) put_block_header(rec + 32 + (size_t)4 * N, N, q->dk_all[q->off[t] + j], (float)ls);
        // beta = g / emission of this site; the row words are fetched again (the registers hold the next row's by now)
        {
          using RR = RowRounds<MPL>;
          const double inv_match = q->p.c.inv_ntheta_d, inv_mis = q->p.c.inv_theta_d;
          swait();
          static_for<0, RR::NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            constexpr int w0 = RR::first(r), nw = RR::size(r);
            const uint64_t* rpl = rp;
            sload_round<w0, nw>(rpl);
            swait();
            static_for<0, nw>([&](auto ic) __attribute__((always_inline)) {
              constexpr int i = decltype(ic)::value;
              const double im = inv_match, ix = inv_mis;
              const uint64_t skl = skm;
              const uint32_t al = lds_mine;
              state_unemitted_to_lds<w0 + i, kBufA + 2 * i>(al, im, ix, skl);
            });
          });
          swait();
          __syncthreads();
          float* dst = reinterpret_cast<float*>(rec + 56 + (size_t)4 * N);
          for (int n = threadIdx.x; n < N; n += 64 * G) dst[n] = stage[n];
          __syncthreads();
          const uint64_t* rpn = rp_next;
          sload_round<0, RR::size(0)>(rpn);  // what the next sweep expects there
        }
        wcur--;
        *wslot = wcur;
        stone_j = wcur >= 0 ? __builtin_amdgcn_readfirstlane(q->bej_all[(long long)t * q->p.W + wcur]) : -1;
      }
      rp = rp_next;
      rp_next = rp_next2;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    finish();
    return;
  }

  // ---------------- backward, two emission probabilities (FastPainting(N, theta) beside Data::theta) ----------------
  else {
    const double theta_d = ka.p.c.theta_d, ntheta_d = ka.p.c.ntheta_d;
    double ls = ka.p.c.log_Nm1 - (double)nd * ka.p.c.log_ntheta;
    const uint64_t* row_last = rows_w + (size_t)(ka.p.L - 1) * stride;
    double acc = 0.0;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int wi = wbase + i;
      const uint64_t w = row_last[i];
      const bool v = __builtin_amdgcn_inverse_ballot_w64(valid_mask(wi, ka.p.N, k));
      const bool m = __builtin_amdgcn_inverse_ballot_w64(~w & sk_last);
      state_set<i>(1.0);
      acc += v ? (m ? theta_d : ntheta_d) : 0.0;
    });
    double B = block_sum_regs<G>(red, slot, acc);
    *wslot = ka.p.W - 1;
    int stone_j = nd - 1;  // the last window ends at the last SNP
    // stones taken at the last SNP hold beta = 1 for every state, the target included (:427-432,:447-462)
    while (stone_j == nd - 1) {
      KArgsPtr q = rare_args();
      const int N = q->p.N;
      int wcur = next_window();
      uint8_t* rec = q->out + ((size_t)wcur * q->p.K + t) * (56 + (size_t)8 * N);
      if (first_thread) put_block_header(rec + 32 + (size_t)4 * N, N, q->p.L - 1, (float)ls);
      float* dst = reinterpret_cast<float*>(rec + 56 + (size_t)4 * N);
      for (int n = threadIdx.x; n < N; n += 64 * G) dst[n] = 1.0f;
      wcur--;
      *wslot = wcur;
      stone_j = wcur >= 0 ? __builtin_amdgcn_readfirstlane(q->bej_all[(long long)t * q->p.W + wcur]) : -1;
    }
    stateless();
    set_rows_base(rows_w);
    set_edge_mask(sk_first);
    uint32_t stride_bytes = 8u * (uint32_t)stride;
    asm volatile("" : "+s"(stride_bytes));
    set_touch_lane(4u * (uint32_t)lane);
    // constants of the update in vector registers, opaque to the compiler: fetched again from the argument
    // segment inside the loop they would bring a wait with them
    double rho = ka.p.c.rho_theta;
    asm volatile("" : "+v"(rho));
    const double inv_nt = sgpr_f64(ka.p.c.inv_ntheta_d), inv_t = sgpr_f64(ka.p.c.inv_theta_d);
    const ScalarWeight w_all = {sgpr_f64(ntheta_d)}, w_mis = {sgpr_f64(theta_d - ntheta_d)};
    double coef_j = sgpr_f64(rec[nd - 1].coef), nx_j = sgpr_f64(rec[nd - 1].nx);  // terms used by the step to site j: index j + 1
    set_site_pointer(rec + (nd - 2));
    const uint64_t* rp_n = row_pointer(ka.p.L - 1, stride_bytes);
    const uint64_t* rp_c = row_pointer(__builtin_amdgcn_readfirstlane(dk[nd - 2]), stride_bytes);
    const uint64_t* rp_p = row_pointer(__builtin_amdgcn_readfirstlane(dk[nd > 2 ? nd - 3 : 0]), stride_bytes);
    sload_row2<kBufA, 0, RowRounds2<MPL>::size(0)>(rp_n);
    sload_row2<kBufC, 0, RowRounds2<MPL>::size(0)>(rp_c);
    uint32_t wr_addr = 0, rd_base = 0;
    if constexpr (G > 1) {
      exchange_addresses<G>(red, wr_addr, rd_base);
      __syncthreads();  // every wave has read the entries of the sum above
    }
    uint64_t skn = sk_last;  // the row handled before the first step is the last SNP's; every later one is a site of the target
    for (int j = nd - 2; j >= 0; j--) {
      if ((j & 31) == 0) rotate_priority(rank, j, 1);
      // beta' = (beta + mis'*b + a) * (mis' ? rho : 1), a = rx/(1-theta), b = rx/theta - a (:488-502), as
      // t = beta + a for every state, then fma(t, rho, b*rho) for the mismatching ones
      const double rx = B * coef_j;
      const double aa = rx * inv_nt;
      const double bb = rx * inv_t - aa;
      const double c1 = bb * rho;
      write_stateless<MPL, true>(rx, inv_nt);
      // scalars of the step to site j-1 travel during this sweep
      prefetch_site<false>();
      const uint64_t skc = edge_mask(j, 0);
      ls += nx_j;
      double a0, a1, m0 = 0.0, m1 = 0.0;
      sweep_two_rows<MPL>(a0, a1, m0, m1, aa, rho, c1, skn, skc, rp_n, rp_c, rp_p);
      // B = (1-theta) * sum over matching states + theta * sum over mismatching states (:509-517)
      uint32_t B_high;
      if constexpr (G == 1) {
        B = block_sum_hi<G>(red, slot, B_high, a0 + a1, w_all, m0 + m1, w_mis);
      } else {
        B = exchange_sum<G>(wr_addr, rd_base, j, a0 + a1, w_all, m0 + m1, w_mis);
        B_high = sgpr32((uint32_t)__double2hiint(B));
      }
      int s_prev2;
      fetched_site<false>(s_prev2, coef_j, nx_j);
      const uint64_t* rp_p2 = row_pointer(s_prev2, stride_bytes);
      if constexpr (kTouchRows) touch_row<MPL>(rp_p2);
      if (!surely_in_range(B_high) && out_of_range_exact(B)) {
        const double inv = 1.0 / B;
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
          const double f = inv;
          state_scale<decltype(ic)::value>(f);
        });
        ls += (double)fast_log_dev((float)B);  // float approximation on this side (:562)
        B = 1.0;
      }
      while (stone_j == j) {
        KArgsPtr q = rare_args();
        const int N = q->p.N;
        int wcur = next_window();
        uint8_t* rec = q->out + ((size_t)wcur * q->p.K + t) * (56 + (size_t)8 * N);
        if (first_thread) put_block_header(rec + 32 + (size_t)4 * N, N, q->dk_all[q->off[t] + j], (float)ls);
        store_states(reinterpret_cast<float*>(rec + 56 + (size_t)4 * N), N);
        wcur--;
        *wslot = wcur;
        stone_j = wcur >= 0 ? __builtin_amdgcn_readfirstlane(q->bej_all[(long long)t * q->p.W + wcur]) : -1;
      }
      rp_n = rp_c;
      rp_c = rp_p;
      rp_p = rp_p2;
      skn = ~0ull;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  finish();
}

template <int MPL, int G>
inline hipError_t launch_regs_t(const PaintArgs& a, hipStream_t st)
{
  // the stone staging buffer must fit in the 160 KiB of LDS of a CU
  constexpr size_t lds_bytes = (size_t)G * MPL * 256 + 3 * G * 8;
  if constexpr (lds_bytes > 160 * 1024) {
    return hipErrorInvalidValue;
  } else {
    if (lds_bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&paint_kernel_r<MPL, G>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return e;
    }
    KArgs ka;
    ka.rows = a.rows;
    ka.dk_all = a.dk;
    ka.rec_all = a.rec;
    ka.off = a.off;
    ka.nd_all = a.nd;
    ka.bbj_all = a.bbj;
    ka.bej_all = a.bej;
    ka.order = a.order;
    ka.wb = a.wb;
    ka.out = a.out;
    ka.trace = a.trace;
    ka.p.row_stride = a.row_stride;
    ka.p.N = a.N;
    ka.p.L = a.L;
    ka.p.W = a.W;
    ka.p.k_begin = a.k_begin;
    ka.p.K = a.K;
    ka.p.c = a.c;
    hipLaunchKernelGGL((paint_kernel_r<MPL, G>), dim3(2 * (unsigned)a.K), dim3(64 * G), lds_bytes, st, ka);
    return hipGetLastError();
  }
}

// Dispatch on MPL for one G.
template <int G>
inline hipError_t launch_regs_g(const PaintArgs& a, int MPL, hipStream_t st)
{
  switch (MPL) {
#define RELATE_CASE(M) \
  case M: return launch_regs_t<M, G>(a, st);
    RELATE_CASE(4) RELATE_CASE(8) RELATE_CASE(12) RELATE_CASE(16) RELATE_CASE(20) RELATE_CASE(24) RELATE_CASE(28)
    RELATE_CASE(32) RELATE_CASE(36) RELATE_CASE(40) RELATE_CASE(48) RELATE_CASE(56) RELATE_CASE(64) RELATE_CASE(72)
    RELATE_CASE(80)
#undef RELATE_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace regs
}  // namespace relate
