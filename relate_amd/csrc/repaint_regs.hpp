// Re-painting of one window from its stepping stones (FastPainting::RePaintSection,
// src/fast_painting.cpp:654-1126) for all prepared targets, with the register-resident sweeps of
// paint_regs.hpp: the section of target t is the piece [bbj, bej] of its derived-site list, seeded with the
// alpha stone at its first site and the beta stone at its last, and the product alpha * beta of every site
// of the section is written as a row of the `topology` matrix (float), with its logscale.
//
// Two launches per window.  The forward launch keeps every alpha row (double, :786-919 keeps them in
// `alpha[j]`) in a scratch matrix in HBM; the backward launch reads them back while it walks the section
// in the other direction and writes float(alpha[j][n] * beta[j][n]) (:958-969,:1067-1078), taken before the
// beta row is rescaled (:1081-1095).  Rows have a pitch of G*MPL*64 elements (>= N), so the lanes past N
// and the target's own lane, which hold exact zeros, are stored like any other.
//
// Internal header; instantiated per G by paint_regs_g*.hip.
#pragma once
#include "paint_regs.hpp"

namespace relate {
namespace regs {

struct RepaintKArgs {
  const uint64_t* rows;
  const int32_t* dk_all;
  const SiteRec* rec_all;
  const int64_t* off;
  const int32_t* nd_all;
  const int32_t* bbj_all;   // [K][W] list index of the alpha stone of window w
  const int32_t* bej_all;   // [K][W] list index of the beta stone
  const int32_t* order;     // [K] launch order
  const uint8_t* stones;    // [K] records of this window (56+8N bytes each), target t at t*(56+8N)
  const int64_t* row_off;   // [K] first row of target t in scratch / topology / logscales
  const double* last_coef;  // [K] gap terms of the section's last site: r[last] alone (:737-750)
  const double* last_nx;
  double* scratch;          // [rows][pitch] alpha rows
  float* topology;          // [rows][pitch]
  float* logscales;         // [rows]
  int w, pitch;
  Scal p;
};
// Pointers that went through an empty asm (to keep the compiler from fetching them again inside a loop) have
// lost their provenance; without it the accesses become FLAT instructions, whose waits include the scalar-load
// counter and with it the row fetch in flight.  These casts say "global memory" again.
typedef __attribute__((address_space(1))) double GlobalDouble;
typedef __attribute__((address_space(1))) float GlobalFloat;
__device__ __forceinline__ GlobalDouble* as_global(double* p) { return (GlobalDouble*)p; }
__device__ __forceinline__ const GlobalDouble* as_global(const double* p) { return (const GlobalDouble*)p; }
__device__ __forceinline__ GlobalFloat* as_global(float* p) { return (GlobalFloat*)p; }

template <int I>
__device__ __forceinline__ double state_get()
{
  double v;
  asm volatile("v_mov_b64 %0, v[%1:%1+1]" : "=v"(v) : "n"(kA0 + 2 * I));
  return v;
}

template <int MPL, int G, bool BWD>
__global__ void RELATE_REGS_ATTR __launch_bounds__(64 * G) repaint_kernel_r(const RepaintKArgs ka)
{
  static_assert(MPL >= 4 && MPL <= kMaxMPL, "states per lane");
  extern __shared__ double lds_dyn[];
  double (*red)[G] = reinterpret_cast<double (*)[G]>(lds_dyn);
  claim_registers<MPL>();
  int slot = 0;
  const int rank = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 3;
  const int t = __builtin_amdgcn_readfirstlane(ka.order[blockIdx.x]);
  const int k = ka.p.k_begin + t;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wbase = wave * MPL;
  const int stride = ka.p.row_stride;
  const int N = ka.p.N;
  const long long o = ka.off[t];
  const int32_t* __restrict__ dk = ka.dk_all + o;
  const SiteRec* __restrict__ rec = ka.rec_all + o;
  const int nd_chunk = __builtin_amdgcn_readfirstlane(ka.nd_all[t]);
  const int jb = __builtin_amdgcn_readfirstlane(ka.bbj_all[(long long)t * ka.p.W + ka.w]);
  const int je = __builtin_amdgcn_readfirstlane(ka.bej_all[(long long)t * ka.p.W + ka.w]);
  const int nds = je - jb + 1;  // sites of the section (>= 2: the two stones sit on different sites)
  const int kw = k >> 6, kb = k & 63;
  const uint64_t* rows_w = ka.rows + wbase;
  const size_t R = 56 + (size_t)8 * N;
  const uint8_t* stone = ka.stones + (size_t)t * R;
  const size_t row0 = (size_t)ka.row_off[t];
  const size_t pitch = (size_t)ka.pitch;
  const size_t col = (size_t)wbase * 64 + lane;

  auto stateless = [&]() __attribute__((always_inline)) {
    const int full_words = N >> 6, rem = N & 63;
    const bool self_here = kw >= wbase && kw < wbase + MPL;
    const int w_ph = full_words > wbase ? full_words : wbase;
    set_stateless(sgpr64(self_here ? (1ull << kb) : 0ull), sgpr32(self_here ? 2u * (uint32_t)(kw - wbase) : 0u),
                  sgpr64((w_ph == full_words && rem) ? ~((1ull << rem) - 1ull) : ~0ull),
                  sgpr32(w_ph - wbase < MPL ? 2u * (uint32_t)(w_ph - wbase) : 2u * MPL), 2u * MPL);
  };

  const int s_first = __builtin_amdgcn_readfirstlane(dk[jb]);
  const int s_last = __builtin_amdgcn_readfirstlane(dk[je]);
  // the target's own allele at the two ends of the section; it is '1' at every site in between
  const uint64_t sk_first = sgpr64(((ka.rows[(size_t)s_first * stride + kw] >> kb) & 1ull) ? ~0ull : 0ull);
  const uint64_t sk_last = sgpr64(((ka.rows[(size_t)s_last * stride + kw] >> kb) & 1ull) ? ~0ull : 0ull);
  const double rho = ka.p.c.rho_theta;

  if constexpr (!BWD) {
    // ---------------- forward (:786-919) ----------------
    const float* __restrict__ alpha0 = reinterpret_cast<const float*>(stone + 32);
    double prev = (double)*reinterpret_cast<const float*>(stone + 8 + 20);  // logscale of the alpha stone
    double acc = 0.0;
    double* arow = ka.scratch + row0 * pitch + col;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int n = (wbase + i) * 64 + lane;
      const double a = (n < N && n != k) ? (double)alpha0[n] : 0.0;
      state_set<i>(a);
      arow[(size_t)i * 64] = a;
      acc += a;
    });
    double S = block_sum_regs<G>(red, slot, acc);
    if (threadIdx.x == 0) ka.logscales[row0] = (float)prev;
    stateless();
    set_thresholds(ka.p.c.lo, ka.p.c.hi);
    int s_cur = __builtin_amdgcn_readfirstlane(dk[jb + 1]);
    int s_next = __builtin_amdgcn_readfirstlane(dk[jb + 2 < nd_chunk ? jb + 2 : nd_chunk - 1]);
    double coef_j = rec[jb].coef, nx_j = rec[jb].nx;
    set_site_pointer(rec + jb + 1);
    // what every site needs from the argument segment, fetched once and opaque to the compiler: a scalar load
    // inside the loop would bring a wait for the row fetch in flight with it
    double* scratch_rows = ka.scratch + row0 * pitch + col;
    float* ls_rows = ka.logscales + row0;
    size_t pitch_l = pitch;
    int stride_l = stride;
    asm volatile("" : "+s"(pitch_l), "+s"(stride_l), "+s"(ls_rows));
    asm volatile("" : "+v"(scratch_rows));
    double rho_l = rho;
    asm volatile("" : "+v"(rho_l));
    const uint64_t* rp = rows_w + (size_t)s_cur * stride_l;
    sload_round<0, RowRounds<MPL>::size(0)>(rp);
    for (int j = 1; j < nds; j++) {
      if ((j & 31) == 0) rotate_priority(rank, j, 0);
      const double rx = S * coef_j;
      write_stateless<MPL, false>(S, coef_j);
      prefetch_site<true>();
      const uint64_t skm = edge_select(j, nds - 1, sk_last);
      prev += nx_j;
      double c0, c1;
      const uint64_t* rp_next = rows_w + (size_t)s_next * stride_l;
      sweep_row<MPL>(c0, c1, rx, rho_l, skm, rp, rp_next);
      S = block_sum_regs<G>(red, slot, c0, c1);
      int s_next2;
      fetched_site<true>(s_next2, coef_j, nx_j);
      float lsj = (float)prev;  // (:839-841) the float is what the rescale term is added to
      if (out_of_range(S)) {
        const double inv = 1.0 / S;
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
          const double f = inv;
          state_scale<decltype(ic)::value>(f);
        });
        const double lg = log(S);
        prev += lg;
        lsj = (float)((double)lsj + lg);
        S = 1.0;
      }
      {
        if (threadIdx.x == 0) as_global(ls_rows)[j] = lsj;
        GlobalDouble* __restrict__ ar = as_global(scratch_rows + (size_t)j * pitch_l);
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          ar[(size_t)i * 64] = state_get<i>();
        });
      }
      s_cur = s_next;
      s_next = s_next2;
      rp = rp_next;
    }
    swait();
  } else {
    // ---------------- backward (:921-1107) ----------------
    const double theta_d = ka.p.c.theta_d, ntheta_d = ka.p.c.ntheta_d;
    const float* __restrict__ beta0 = reinterpret_cast<const float*>(stone + 56 + (size_t)4 * N);
    const float ls_beta = *reinterpret_cast<const float*>(stone + 32 + (size_t)4 * N + 20);
    double prev = (double)ls_beta;
    const uint64_t* row_last = rows_w + (size_t)s_last * stride;
    double acc = 0.0;
    {
      const double* arow = ka.scratch + (row0 + nds - 1) * pitch + col;
      float* trow = ka.topology + (row0 + nds - 1) * pitch + col;
      static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const int n = (wbase + i) * 64 + lane;
        const uint64_t w = row_last[i];
        const bool m = __builtin_amdgcn_inverse_ballot_w64(~w & sk_last);
        const double b = (n < N && n != k) ? (double)beta0[n] : 0.0;
        state_set<i>(b);
        acc += (m ? theta_d : ntheta_d) * b;
        trow[(size_t)i * 64] = (float)(arow[(size_t)i * 64] * b);
      });
    }
    double B = block_sum_regs<G>(red, slot, acc);
    if (threadIdx.x == 0) ka.logscales[row0 + nds - 1] = ka.logscales[row0 + nds - 1] + ls_beta;  // float + float (:929)
    stateless();
    set_thresholds(ka.p.c.lo, ka.p.c.hi);
    const double inv_nt = ka.p.c.inv_ntheta_d, inv_t = ka.p.c.inv_theta_d;
    const double w_all = ntheta_d, w_mis = theta_d - ntheta_d;
    int s_cur = __builtin_amdgcn_readfirstlane(dk[je - 1]);
    int s_prev = __builtin_amdgcn_readfirstlane(dk[je > 1 ? je - 2 : 0]);
    double coef_j = ka.last_coef[t], nx_j = ka.last_nx[t];
    set_site_pointer(rec + (je - 1));
    const double* scratch_rows = ka.scratch + row0 * pitch + col;
    float* top_rows = ka.topology + row0 * pitch + col;
    float* ls_rows = ka.logscales + row0;
    size_t pitch_l = pitch;
    int stride_l = stride;
    asm volatile("" : "+s"(pitch_l), "+s"(stride_l), "+s"(ls_rows));
    asm volatile("" : "+v"(scratch_rows), "+v"(top_rows));
    double rho_l = rho;
    asm volatile("" : "+v"(rho_l));
    const double inv_nt_s = sgpr_f64(inv_nt), inv_t_s = sgpr_f64(inv_t);
    const ScalarWeight w_all_s = {sgpr_f64(w_all)}, w_mis_s = {sgpr_f64(w_mis)};
    const uint64_t* rp_n = row_last;
    const uint64_t* rp_c = rows_w + (size_t)s_cur * stride_l;
    sload_words<kBufA, 0, RowRounds2<MPL>::size(0)>(rp_n);
    sload_words<kBufC, 0, RowRounds2<MPL>::size(0)>(rp_c);
    for (int j = nds - 2; j >= 0; j--) {
      if ((j & 31) == 0) rotate_priority(rank, j, 0);
      const double rx = B * coef_j;
      const double aa = rx * inv_nt_s;
      const double bb = rx * inv_t_s - aa;
      const double c1 = bb * rho_l;
      write_stateless<MPL, true>(rx, inv_nt_s);
      prefetch_site<false>();
      const uint64_t skn = edge_select(j, nds - 2, sk_last);
      const uint64_t skc = edge_select(j, 0, sk_first);
      prev += nx_j;
      const float ls_fwd = as_global(ls_rows)[j];  // the forward launch's logscale of this row; asked for early
      double a0, a1, m0 = 0.0, m1 = 0.0;
      const uint64_t* rp_p = rows_w + (size_t)s_prev * stride_l;
      sweep_two_rows<MPL>(a0, a1, m0, m1, aa, rho_l, c1, skn, skc, rp_n, rp_c, rp_p);
      uint32_t B_high;
      B = block_sum_hi<G>(red, slot, B_high, a0 + a1, w_all_s, m0 + m1, w_mis_s);
      int s_prev2;
      fetched_site<false>(s_prev2, coef_j, nx_j);
      {
        // topology row j = alpha row j * beta row j, before the beta row is rescaled (:1067-1095)
        // the loads of the whole row go out before the first store: the two rows never overlap
        const GlobalDouble* __restrict__ ar = as_global(scratch_rows + (size_t)j * pitch_l);
        GlobalFloat* __restrict__ tr = as_global(top_rows + (size_t)j * pitch_l);
        // in groups of 8 words: all loads of a group, then its products and stores (left alone, the compiler
        // keeps two loads in flight and pays a memory round trip per word)
        constexpr int CH = 8;
        static_for<0, (MPL + CH - 1) / CH>([&](auto gc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value;
          constexpr int n = MPL - CH * g < CH ? MPL - CH * g : CH;
          double av[n];
          static_for<0, n>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            av[i] = ar[(size_t)(CH * g + i) * 64];
          });
          __builtin_amdgcn_sched_barrier(0);
          static_for<0, n>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            tr[(size_t)(CH * g + i) * 64] = (float)(av[i] * state_get<CH * g + i>());
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      float lsj = (float)((double)ls_fwd + prev);  // (:994-997)
      if (out_of_range(B)) {
        const double inv = 1.0 / B;
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
          const double f = inv;
          state_scale<decltype(ic)::value>(f);
        });
        const double lg = log(B);  // std::log on this side too (:1091-1092)
        prev += lg;
        lsj = (float)((double)lsj + lg);
        B = 1.0;
      }
      if (threadIdx.x == 0) as_global(ls_rows)[j] = lsj;
      s_cur = s_prev;
      s_prev = s_prev2;
      rp_n = rp_c;
      rp_c = rp_p;
    }
    swait();
  }
}

typedef RepaintLaunch RepaintArgs;  // paint_kernels.h

template <int MPL, int G>
inline hipError_t launch_repaint_t(const RepaintArgs& a, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)2 * G * 8;
  RepaintKArgs ka;
  ka.rows = a.rows;
  ka.dk_all = a.dk;
  ka.rec_all = a.rec;
  ka.off = a.off;
  ka.nd_all = a.nd;
  ka.bbj_all = a.bbj;
  ka.bej_all = a.bej;
  ka.order = a.order;
  ka.stones = a.stones;
  ka.row_off = a.row_off;
  ka.last_coef = a.last_coef;
  ka.last_nx = a.last_nx;
  ka.scratch = a.scratch;
  ka.topology = a.topology;
  ka.logscales = a.logscales;
  ka.w = a.w;
  ka.pitch = a.pitch;
  ka.p.row_stride = a.row_stride;
  ka.p.N = a.N;
  ka.p.L = a.L;
  ka.p.W = a.W;
  ka.p.k_begin = a.k_begin;
  ka.p.K = a.K;
  ka.p.c = a.c;
  if (a.pitch != G * MPL * 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL((repaint_kernel_r<MPL, G, false>), dim3((unsigned)a.K), dim3(64 * G), lds_bytes, st, ka);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((repaint_kernel_r<MPL, G, true>), dim3((unsigned)a.K), dim3(64 * G), lds_bytes, st, ka);
  return hipGetLastError();
}

template <int G>
inline hipError_t launch_repaint_g(const RepaintArgs& a, int MPL, hipStream_t st)
{
  switch (MPL) {
#define RELATE_CASE(M) \
  case M: return launch_repaint_t<M, G>(a, st);
    RELATE_CASE(4) RELATE_CASE(8) RELATE_CASE(12) RELATE_CASE(16) RELATE_CASE(20) RELATE_CASE(24) RELATE_CASE(28)
    RELATE_CASE(32) RELATE_CASE(36) RELATE_CASE(40) RELATE_CASE(48) RELATE_CASE(56) RELATE_CASE(64) RELATE_CASE(72)
    RELATE_CASE(80)
#undef RELATE_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace regs
}  // namespace relate
