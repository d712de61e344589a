// Painting recursion of one target and pass per workgroup with the steady-state loop in hand-scheduled gfx950
// assembly (paint_loop_gen.hpp, written by tools/gen_paint_loop.py): forward pass, and the backward pass on the
// emission-weighted state, which obeys the forward recursion when painter and data hold one theta
// (FastPainting(Data&), src/fast_painting.hpp:32-45).  Two thetas take the kernel of paint_regs.hpp.
//
// Division of labour.  Everything a site needs from the previous one lives in registers that are addressed by
// number: the MPL state words of a lane in v[kA0 ..], the sum, the running logscale, the prefetched row words
// and gap terms, the row pointers.  The loop (one asm statement) runs site after site without a single
// compiler-generated instruction and returns to C++ only for what is rare: a sum outside the band that is
// surely inside the rescale thresholds, a stone that is due, the end of the list.  The C++ around it reads and
// writes those registers through the small asm accessors below; the compiler itself keeps to v0 .. v[kCV-1] and
// s0 .. s[kCS-1] (amdgpu_num_vgpr / amdgpu_num_sgpr; tools/check_regs.py reads the assembly of every build), so
// whatever it spills, it spills in the rare paths.
//
// Internal header; instantiated per G by paint_loop_g*.hip.
#pragma once
#include "paint_loop_gen.hpp"
#include "paint_regs.hpp"

namespace relate {
namespace aloop {

using namespace dev;
using regs::KArgs;
using regs::KArgsPtr;
using regs::rare_args;
using regs::sgpr32;
using regs::sgpr64;
using regs::sgpr_f64;
using regs::static_for;

// amdgpu_num_vgpr is doubled by the gfx90a+ back end (unified VGPR/AGPR file; no AGPRs are used here);
// amdgpu_num_sgpr counts VCC, FLAT_SCRATCH and XNACK_MASK on top of s0 .. s[kCS-1].
#define RELATE_A_VGPR_ATTR 12
#define RELATE_A_SGPR_ATTR 32
static_assert(2 * RELATE_A_VGPR_ATTR == kCV && RELATE_A_SGPR_ATTR == kCS + 8, "register limits against the generated map");
#define RELATE_A_ATTR __attribute__((amdgpu_num_vgpr(RELATE_A_VGPR_ATTR), amdgpu_num_sgpr(RELATE_A_SGPR_ATTR)))

// ---- accessors of the fixed registers -------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void set_s32(uint32_t v)
{
  asm volatile("s_mov_b32 s[%1], %0" ::"s"(v), "n"(R));
}
template <int R>
__device__ __forceinline__ void set_s64(uint64_t v)
{
  asm volatile("s_mov_b64 s[%1:%1+1], %0" ::"s"(v), "n"(R));
}
template <int R>
__device__ __forceinline__ uint64_t get_s64()
{
  uint64_t v;
  asm volatile("s_mov_b64 %0, s[%1:%1+1]" : "=s"(v) : "n"(R));
  return v;
}
template <int R>
__device__ __forceinline__ void set_v32(uint32_t v)
{
  asm volatile("v_mov_b32 v[%1], %0" ::"v"(v), "n"(R));
}
template <int R>
__device__ __forceinline__ void set_v64(double v)
{
  asm volatile("v_mov_b64 v[%1:%1+1], %0" ::"v"(v), "n"(R));
}
template <int R>
__device__ __forceinline__ double get_v64()
{
  double v;
  asm volatile("v_mov_b64 %0, v[%1:%1+1]" : "=v"(v) : "n"(R));
  return v;
}
__device__ __forceinline__ void swait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

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
               "n"(kV_T + (I & 1)), "n"(256 * I)
               : "memory");
}
// float(state I * (inv_mis where the lane mismatches row word s[SW:SW+1] under the current site's own-allele mask
// s[kS_SKM ..], else inv_match)) -> LDS
template <int I, int SW>
__device__ __forceinline__ void state_unemitted_to_lds(uint32_t lds_addr, double inv_match, double inv_mis)
{
  asm volatile(
      "v_mul_f64 v[%[t]:%[t]+1], v[%[a]:%[a]+1], %[im]\n\t"
      "s_andn2_b64 exec, s[%[sk]:%[sk]+1], s[%[w]:%[w]+1]\n\t"
      "v_mul_f64 v[%[t]:%[t]+1], v[%[a]:%[a]+1], %[ix]\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_cvt_f32_f64 v[%[x]], v[%[t]:%[t]+1]\n\t"
      "ds_write_b32 %[addr], v[%[x]] offset:%[off]" ::[addr] "v"(lds_addr),
      [im] "v"(inv_match), [ix] "v"(inv_mis), [sk] "n"(kS_SKM), [a] "n"(kA0 + 2 * I), [t] "n"(kV_T), [x] "n"(kV_X + (I & 1)),
      [w] "n"(SW), [off] "n"(256 * I)
      : "memory", "scc");
}
// The record at s[kS_REC] into its landing registers, and the pointer on to the next one of the pass: what the
// loop expects to be on its way when it is entered.
__device__ __forceinline__ void fetch_record()
{
  asm volatile(
      "s_load_dwordx4 s[%0:%0+3], s[%2:%2+1], 0\n\t"
      "s_load_dwordx2 s[%1:%1+1], s[%2:%2+1], 16\n\t"
      "s_add_u32 s[%2], s[%2], s[%3]\n\t"
      "s_addc_u32 s[%2+1], s[%2+1], s[%3+1]" ::"n"(kS_RECBUF),
      "n"(kS_DKF), "n"(kS_REC), "n"(kS_RECSTEP)
      : "memory", "scc");
}

template <int MPL, int G>
__global__ void RELATE_A_ATTR __launch_bounds__(64 * G) paint_kernel_a(const KArgs ka)
{
  static_assert(MPL >= 4 && kA0 + 2 * MPL <= 256, "states per lane");
  // Dynamic LDS: two rows of G partial sums (the loop's exchange between the waves of a target), G window
  // counters, then the staging buffer of one stone payload (G * MPL * 64 floats).
  extern __shared__ double lds_dyn[];
  double (*red)[G] = reinterpret_cast<double (*)[G]>(lds_dyn);
  volatile int* wslot = reinterpret_cast<volatile int*>(lds_dyn + 2 * G) + 2 * (threadIdx.x >> 6);
  auto next_window = [&]() __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*wslot); };
  float* stage = reinterpret_cast<float*>(lds_dyn + 3 * G);
  Claim<MPL>::all();
  // The two passes of a target are independent until their stones meet in the record: two workgroups.  Which
  // of the pair 2i, 2i+1 takes which pass follows the parity of the set bits of the block index, so that every
  // power-of-two stride of the dispatcher (XCDs, CUs, SIMDs) sees both kinds.
  const int pass = __builtin_popcount(blockIdx.x) & 1;
  const uint64_t trace_t0 = __builtin_amdgcn_s_memrealtime();
  // slot of this wave on its SIMD (HW_ID.wave_id) sets the phase of its priority rotation; the waves of a
  // multi-wave workgroup meet at a barrier every site and must share one schedule: they take it from the block index
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
  const uint32_t lds_mine = (uint32_t)(uintptr_t)(stage + wbase * 64 + lane);

  // bit of the target itself at the two forced sites; it is '1' at every other site of its list
  const uint64_t sk_first = sgpr64(((ka.rows[kw] >> kb) & 1ull) ? ~0ull : 0ull);
  const uint64_t sk_last = sgpr64(((ka.rows[(size_t)(ka.p.L - 1) * stride + kw] >> kb) & 1ull) ? ~0ull : 0ull);

  // What both passes set up the same way.  Lanes without a state: the target's own (its bit, or 0 if its word
  // belongs to another wave) and the padding lanes from the first incomplete word of this wave on; they never
  // mismatch (the target carries the derived allele at its own sites, pack_rows sets the bits past N), so the
  // sweep only adds rx to them: loaded with -rx before every sweep they come out as exact zeros.
  auto common_registers = [&]() __attribute__((always_inline)) {
    const int N = ka.p.N;
    const int full_words = N >> 6, rem = N & 63;
    const bool self_here = kw >= wbase && kw < wbase + MPL;
    const int w_ph = full_words > wbase ? full_words : wbase;
    const bool pad_here = w_ph - wbase < MPL;
    set_s64<kS_SELFM>(sgpr64(self_here ? (1ull << kb) : 0ull));
    set_s32<kS_SELFIDX>(sgpr32(self_here ? 2u * (uint32_t)(kw - wbase) : 0u));
    set_s64<kS_PHM>(sgpr64((w_ph == full_words && rem) ? ~((1ull << rem) - 1ull) : ~0ull));
    set_s32<kS_PHIDX>(sgpr32(pad_here ? 2u * (uint32_t)(w_ph - wbase) : 2u * MPL));
    set_s32<kS_HASSL>(sgpr32((self_here || pad_here) ? 1u : 0u));
    set_s64<kS_ROWSW>(sgpr64((uint64_t)(uintptr_t)rows_w));
    set_s32<kS_STRIDE>(sgpr32(8u * (uint32_t)stride));
    set_s32<kS_RANK>(sgpr32((uint32_t)rank));
    set_v64<kV_RHO>(ka.p.c.rho_theta);
    set_v32<kV_LANEB>(4u * (uint32_t)lane);
    if constexpr (G > 1) {
      // the exchange starts with the second of its two LDS rows: the first has just served the sum of the
      // initial states, which a slow wave may still be reading
      const uint32_t base = (uint32_t)(uintptr_t)lds_dyn + 8u * G;
      set_v32<kV_LDSR>(base);
      set_v32<kV_LDSW>(base + 8u * (uint32_t)wave);
    }
  };
  auto row_of = [&](int site) __attribute__((always_inline)) {
    return sgpr64((uint64_t)(uintptr_t)(rows_w + (size_t)site * (size_t)stride));
  };

  // Stone payload (src/fast_painting.cpp:368-388,:573-592): every lane drops its states as floats into LDS, then
  // the workgroup streams the N floats out with coalesced stores; the target's own entry is an exact zero already.
  auto store_states = [&](float* __restrict__ dst, int N) __attribute__((always_inline)) {
    const uint32_t a = lds_mine;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) { state_to_lds<decltype(ic)::value>(a); });
    swait();
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 64 * G) dst[n] = stage[n];
    __syncthreads();
  };
  auto finish = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    KArgsPtr q = rare_args();
    if (q->trace && threadIdx.x == 0) {
      uint64_t* tr = q->trace + (size_t)4 * blockIdx.x;
      const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      tr[0] = ((uint64_t)xcc << 32) | hw;
      tr[1] = trace_t0;
      tr[2] = __builtin_amdgcn_s_memrealtime();
      tr[3] = (uint64_t)pass | ((uint64_t)nd << 8);
    }
  };
  // 1 / sum into every state, the sum itself becomes 1
  auto rescale_states = [&](double S) __attribute__((always_inline)) {
    const double inv = 1.0 / S;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
      const double f = inv;
      state_scale<decltype(ic)::value>(f);
    });
    set_v64<kV_S>(1.0);
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
    int slot = 0;
    set_v64<kV_S>(block_sum<G>(acc, red, slot));
    set_v64<kV_LS>(0.0);
    *wslot = 0;
    // The loop counts the sites still to come (left = nd - 1 - j); the next alpha stone is due when `left`
    // reaches stone_left (window 0 starts at site 0).
    int stone_left = nd - 1;
    auto emit = [&](int left) __attribute__((always_inline)) {
      while (stone_left == left) {
        KArgsPtr q = rare_args();
        const int N = q->p.N, W = q->p.W;
        int wcur = next_window();
        uint8_t* out = q->out + ((size_t)wcur * q->p.K + t) * (56 + (size_t)8 * N);
        if (first_thread) {
          put_u32(out, (uint32_t)q->wb[wcur]);
          put_u32(out + 4, (uint32_t)(q->wb[wcur + 1] - 1));
          put_block_header(out + 8, N, q->dk_all[q->off[t] + (q->nd_all[t] - 1 - left)], (float)get_v64<kV_LS>());
        }
        store_states(reinterpret_cast<float*>(out + 32), N);
        wcur++;
        *wslot = wcur;
        stone_left = wcur < W ? __builtin_amdgcn_readfirstlane(q->nd_all[t] - 1 - q->bbj_all[(long long)t * W + wcur]) : -1;
      }
    };
    emit(nd - 1);
    common_registers();
    set_s32<kS_DIR>(0u);
    set_s64<kS_RECSTEP>(24ull);
    set_s64<kS_SKEDGE>(sk_last);
    set_s64<kS_COEF>(sgpr64((uint64_t)__double_as_longlong(rec[0].coef)));  // terms of the gap before site 1
    set_s64<kS_NX>(sgpr64((uint64_t)__double_as_longlong(rec[0].nx)));
    // nd >= 2: sites 0 and L-1 are always listed
    set_s64<kS_RP>(row_of(__builtin_amdgcn_readfirstlane(dk[1])));
    set_s64<kS_RPNEXT>(row_of(__builtin_amdgcn_readfirstlane(dk[2 < nd ? 2 : nd - 1])));
    set_s64<kS_TMP>(get_s64<kS_RPNEXT>());
    set_s64<kS_REC>(sgpr64((uint64_t)(uintptr_t)(rec + 1)));
    set_s32<kS_LEFT>(sgpr32((uint32_t)(nd - 2)));
    set_s32<kS_STONELEFT>(sgpr32((uint32_t)stone_left));
    Rows<MPL>::template of_current<0>();
    fetch_record();
    uint32_t resume = 0;
    for (;;) {
      uint32_t S_high;
      const int left = HotLoop<MPL, G>::run(resume, S_high);
      if (left < 0) break;
      resume = 1;
      if (!regs::surely_in_range(S_high)) {
        const double S = get_v64<kV_S>();
        if (regs::out_of_range_exact(S)) {
          rescale_states(S);
          set_v64<kV_LS>(get_v64<kV_LS>() + log(S));
        }
      }
      emit(left);
      set_s32<kS_STONELEFT>(sgpr32((uint32_t)stone_left));
    }
    finish();
    return;
  }

  // ---------------- backward on g = beta * emission (src/fast_painting.cpp:410-599) ----------------
  // The reference keeps beta and weights it with the emission of the current site when it sums (:509-517).
  // With one theta g obeys the forward recursion, g' = (g + rx) * (mismatch' ? rho : 1) with B = sum g, on the
  // masks of one row; beta itself is needed at the stones only, as g / emission.
  {
    const double theta_d = ka.p.c.theta_d, ntheta_d = ka.p.c.ntheta_d;
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
    int slot = 0;
    set_v64<kV_S>(block_sum<G>(acc, red, slot));
    set_v64<kV_LS>(ka.p.c.log_Nm1 - (double)nd * ka.p.c.log_ntheta);
    *wslot = ka.p.W - 1;
    int stone_j = nd - 1;  // the last window ends at the last SNP
    // stones taken at the last SNP hold beta = 1 for every state, the target included (:427-432,:447-462)
    while (stone_j == nd - 1) {
      KArgsPtr q = rare_args();
      const int N = q->p.N;
      int wcur = next_window();
      uint8_t* out = q->out + ((size_t)wcur * q->p.K + t) * (56 + (size_t)8 * N);
      if (first_thread) put_block_header(out + 32 + (size_t)4 * N, N, q->p.L - 1, (float)get_v64<kV_LS>());
      float* dst = reinterpret_cast<float*>(out + 56 + (size_t)4 * N);
      for (int n = threadIdx.x; n < N; n += 64 * G) dst[n] = 1.0f;
      wcur--;
      *wslot = wcur;
      stone_j = wcur >= 0 ? __builtin_amdgcn_readfirstlane(q->bej_all[(long long)t * q->p.W + wcur]) : -1;
    }
    common_registers();
    set_s32<kS_DIR>(1u);
    set_s64<kS_RECSTEP>(~0ull - 23ull);  // -24
    set_s64<kS_SKEDGE>(sk_first);
    set_s64<kS_COEF>(sgpr64((uint64_t)__double_as_longlong(rec[nd - 1].coef)));  // terms of the step to site j: index j + 1
    set_s64<kS_NX>(sgpr64((uint64_t)__double_as_longlong(rec[nd - 1].nx)));
    set_s64<kS_RP>(row_of(__builtin_amdgcn_readfirstlane(dk[nd - 2])));
    set_s64<kS_RPNEXT>(row_of(__builtin_amdgcn_readfirstlane(dk[nd > 2 ? nd - 3 : 0])));
    set_s64<kS_TMP>(get_s64<kS_RPNEXT>());
    set_s64<kS_REC>(sgpr64((uint64_t)(uintptr_t)(rec + (nd - 2))));
    set_s32<kS_LEFT>(sgpr32((uint32_t)(nd - 2)));
    set_s32<kS_STONELEFT>(sgpr32((uint32_t)stone_j));
    Rows<MPL>::template of_current<0>();
    fetch_record();
    uint32_t resume = 0;
    for (;;) {
      uint32_t B_high;
      const int j = HotLoop<MPL, G>::run(resume, B_high);
      if (j < 0) break;
      resume = 1;
      if (!regs::surely_in_range(B_high)) {
        const double B = get_v64<kV_S>();
        if (regs::out_of_range_exact(B)) {
          rescale_states(B);
          set_v64<kV_LS>(get_v64<kV_LS>() + (double)fast_log_dev((float)B));  // float approximation on this side (:562)
        }
      }
      while (stone_j == j) {
        KArgsPtr q = rare_args();
        const int N = q->p.N;
        int wcur = next_window();
        uint8_t* out = q->out + ((size_t)wcur * q->p.K + t) * (56 + (size_t)8 * N);
        if (first_thread) put_block_header(out + 32 + (size_t)4 * N, N, q->dk_all[q->off[t] + j], (float)get_v64<kV_LS>());
        // beta = g / emission of this site: the row words of the site are fetched again (the registers are
        // receiving the next row's by now, which is asked for once more afterwards)
        {
          using RR = Rows<MPL>;
          const double inv_match = q->p.c.inv_ntheta_d, inv_mis = q->p.c.inv_theta_d;
          swait();
          static_for<0, RR::NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            constexpr int w0 = RR::first(r), nw = RR::size(r);
            RR::template of_current<r>();
            swait();
            static_for<0, nw>([&](auto ic) __attribute__((always_inline)) {
              constexpr int i = decltype(ic)::value;
              const double im = inv_match, ix = inv_mis;
              const uint32_t al = lds_mine;
              state_unemitted_to_lds<w0 + i, kS_BUF + 2 * i>(al, im, ix);
            });
          });
          swait();
          __syncthreads();
          float* dst = reinterpret_cast<float*>(out + 56 + (size_t)4 * N);
          for (int n = threadIdx.x; n < N; n += 64 * G) dst[n] = stage[n];
          __syncthreads();
          RR::first_of_next();
        }
        wcur--;
        *wslot = wcur;
        stone_j = wcur >= 0 ? __builtin_amdgcn_readfirstlane(q->bej_all[(long long)t * q->p.W + wcur]) : -1;
      }
      set_s32<kS_STONELEFT>(sgpr32((uint32_t)stone_j));
    }
    finish();
  }
}

template <int MPL, int G>
inline hipError_t launch_loop_t(const PaintArgs& a, hipStream_t st)
{
  // the stone staging buffer must fit in the 160 KiB of LDS of a CU
  constexpr size_t lds_bytes = (size_t)G * MPL * 256 + 3 * G * 8;
  if constexpr (lds_bytes > 160 * 1024) {
    return hipErrorInvalidValue;
  } else {
    if (lds_bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&paint_kernel_a<MPL, G>),
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
    hipLaunchKernelGGL((paint_kernel_a<MPL, G>), dim3(2 * (unsigned)a.K), dim3(64 * G), lds_bytes, st, ka);
    return hipGetLastError();
  }
}

// Dispatch on MPL for one G.
template <int G>
inline hipError_t launch_loop_g(const PaintArgs& a, int MPL, hipStream_t st)
{
  switch (MPL) {
#define RELATE_CASE(M) \
  case M: return launch_loop_t<M, G>(a, st);
    RELATE_CASE(4) RELATE_CASE(8) RELATE_CASE(12) RELATE_CASE(16) RELATE_CASE(20) RELATE_CASE(24) RELATE_CASE(28)
    RELATE_CASE(32) RELATE_CASE(36) RELATE_CASE(40) RELATE_CASE(48) RELATE_CASE(56) RELATE_CASE(64) RELATE_CASE(72)
    RELATE_CASE(80)
#undef RELATE_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace aloop
}  // namespace relate
