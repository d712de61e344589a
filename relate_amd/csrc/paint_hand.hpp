// Hand-scheduled gfx950 recursion kernel of the stepping-stone painting
// (forward and backward Li-Stephens passes of one target per workgroup).
//
// The update of 64 states is three FP64 vector instructions in the forward pass
// (add, multiply, accumulate) and five in the backward pass; which lanes take
// the mismatch branch is decided by writing the bit-row word (an SGPR pair fed
// by s_load) straight into EXEC, so the emission costs no vector instruction at
// all.  Words travel L2 -> scalar cache -> SGPRs in 4-word chunks
// (s_load_dwordx8), double-buffered: chunk c+1 (or chunk 0 of the next site) is
// in flight while chunk c is consumed.  States that never take part (the
// target itself, lanes past N) are updated like any other but are left out of
// every sum (EXEC = validity mask in the two chunks concerned) and every store.
//
// Every "s" operand of the asm below must really be scalar: hipcc does not
// move a value it chose to keep in VGPRs into SGPRs for an "s" constraint (the
// assembler then rejects the instruction, so a mistake cannot get past the
// build).  Values that reach asm are therefore produced by asm (s_load outputs,
// sgpr64(), edge_select()) and never pass through a select.
//
// Internal header; instantiated per G by paint_hand_g*.hip.
#pragma once
#include <type_traits>

#include "paint_device.hpp"

namespace relate {
namespace hand {

using namespace dev;

typedef uint64_t u64x4 __attribute__((ext_vector_type(4)));

// Compile-time loop: f(std::integral_constant<int, I>) for I in [0, N).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// 4 row words (32 bytes) at p + OFF bytes -> SGPRs; asynchronous until swait().
template <int OFF>
__device__ __forceinline__ void sload4(u64x4& buf, const uint64_t* p)
{
  asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(buf) : "s"(p), "n"(OFF));
}
// Keeps the compiler from hoisting tests on x out of a loop (it would park one flag per unrolled chunk in SGPRs).
__device__ __forceinline__ void opaque(uint32_t& x) { asm volatile("" : "+s"(x)); }
// Ties the buffer to the wait: every later use of `buf` is ordered after it.
__device__ __forceinline__ void swait(u64x4& buf) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(buf)); }

// gfx950 needs one wait state between a vector instruction that writes a VGPR and a v_readfirstlane
// that reads it, and two between a vector instruction that writes an SGPR and a vector instruction
// that reads it (e.g. the v_writelane of a register spill).  The compiler inserts them between its own
// instructions but cannot see into an asm statement, so they are part of it: without the leading
// s_nop the value read is the one the VGPR held before.
__device__ __forceinline__ uint32_t sgpr32(uint32_t x)
{
  uint32_t r;
  asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(r) : "v"(x));
  return r;
}
// A wave-uniform 64-bit value as an SGPR pair.
__device__ __forceinline__ uint64_t sgpr64(uint64_t x)
{
  const uint32_t lo = sgpr32((uint32_t)x), hi = sgpr32((uint32_t)(x >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// (j == j_edge) ? edge_mask : all ones, on the scalar unit.
__device__ __forceinline__ uint64_t edge_select(int j, int j_edge, uint64_t edge_mask)
{
  uint64_t r;
  asm volatile("s_cmp_eq_u32 %1, %2\n\ts_cselect_b64 %0, %3, -1" : "=s"(r) : "s"(j), "s"(j_edge), "s"(edge_mask) : "scc");
  return r;
}

// Pins a value into VGPRs (a "v" operand fed from an SGPR would be copied again at every use).
__device__ __forceinline__ void pin_vgpr(double& x) { asm volatile("" : "+v"(x)); }

// ---- forward: alpha' = (alpha + rx) * (mismatch ? rho : 1)          (src/fast_painting.cpp:302-309)
// skm: all ones where the target is derived at this site (every site of its list but possibly the
// two forced ones), else 0 (then nobody mismatches).  Two words per call.
__device__ __forceinline__ void fwd2_update(double& a0, double& a1, double rx, double rho, uint64_t w0, uint64_t w1,
                                            uint64_t skm)
{
  asm volatile(
      "v_add_f64 %0, %0, %2\n\t"
      "v_add_f64 %1, %1, %2\n\t"
      "s_andn2_b64 exec, %6, %4\n\t"
      "v_mul_f64 %0, %0, %3\n\t"
      "s_andn2_b64 exec, %6, %5\n\t"
      "v_mul_f64 %1, %1, %3\n\t"
      "s_mov_b64 exec, -1\n\t"
      : "+v"(a0), "+v"(a1)
      : "v"(rx), "v"(rho), "s"(w0), "s"(w1), "s"(skm)
      : "scc");
}

// Validity mask of word wi on the scalar unit, left in EXEC:
//   wi == kw ? ms : all ones   (the target's own state)
// & wi == fw ? mt : all ones   (the word where the states run out)
// & wi <  uw ? all ones : 0    (words past the last state)
#define RELATE_VALID_TO_EXEC                     \
  "s_cmp_eq_u32 %[wi], %[kw]\n\t"                \
  "s_cselect_b64 exec, %[ms], -1\n\t"            \
  "s_cmp_eq_u32 %[wi], %[fw]\n\t"                \
  "s_cselect_b64 %[tmp], %[mt], -1\n\t"          \
  "s_and_b64 exec, exec, %[tmp]\n\t"             \
  "s_cmp_lt_u32 %[wi], %[uw]\n\t"                \
  "s_cselect_b64 %[tmp], -1, 0\n\t"              \
  "s_and_b64 exec, exec, %[tmp]\n\t"

struct WordMasks {
  uint64_t ms, mt;     // validity masks of the self word and of the tail word
  uint32_t kw, fw, uw; // their word indices, and the number of words that hold states
};

// acc += a over the states of word wi that exist (self chunk / tail chunk only).
__device__ __forceinline__ void acc_word_masked(double& acc, double a, uint32_t wi, const WordMasks& m)
{
  uint64_t tmp;
  asm volatile(RELATE_VALID_TO_EXEC
               "v_add_f64 %[acc], %[acc], %[a]\n\t"
               "s_mov_b64 exec, -1\n\t"
               : [acc] "+v"(acc), [tmp] "=&s"(tmp)
               : [a] "v"(a), [wi] "s"(wi), [kw] "s"(m.kw), [fw] "s"(m.fw), [uw] "s"(m.uw), [ms] "s"(m.ms), [mt] "s"(m.mt)
               : "scc");
}

// ---- backward: beta' = ((beta + mis'*add_mis) + add_all) * (mis' ? rho : 1) with mis' from the row
// handled one step earlier (wn, skn)                                 (src/fast_painting.cpp:495-502)
__device__ __forceinline__ void bwd2_update(double& b0, double& b1, double add_mis, double add_all, double rho,
                                            uint64_t wn0, uint64_t wn1, uint64_t skn)
{
  asm volatile(
      "s_andn2_b64 exec, %7, %5\n\t"
      "v_add_f64 %0, %0, %2\n\t"
      "s_andn2_b64 exec, %7, %6\n\t"
      "v_add_f64 %1, %1, %2\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_add_f64 %0, %0, %3\n\t"
      "v_add_f64 %1, %1, %3\n\t"
      "s_andn2_b64 exec, %7, %5\n\t"
      "v_mul_f64 %0, %0, %4\n\t"
      "s_andn2_b64 exec, %7, %6\n\t"
      "v_mul_f64 %1, %1, %4\n\t"
      "s_mov_b64 exec, -1\n\t"
      : "+v"(b0), "+v"(b1)
      : "v"(add_mis), "v"(add_all), "v"(rho), "s"(wn0), "s"(wn1), "s"(skn)
      : "scc");
}
// Partial sums of beta' over the current row's (wc, skc) matching (am) and mismatching (ax) states
// (src/fast_painting.cpp:509-517); the weights theta / 1-theta are applied once per site to the sums.
__device__ __forceinline__ void bwd2_acc(double& am0, double& am1, double& ax0, double& ax1, double b0, double b1,
                                         uint64_t wc0, uint64_t wc1, uint64_t skc)
{
  asm volatile(
      "s_andn2_b64 exec, %8, %6\n\t"
      "v_add_f64 %2, %2, %4\n\t"
      "s_orn2_b64 exec, %6, %8\n\t"
      "v_add_f64 %0, %0, %4\n\t"
      "s_andn2_b64 exec, %8, %7\n\t"
      "v_add_f64 %3, %3, %5\n\t"
      "s_orn2_b64 exec, %7, %8\n\t"
      "v_add_f64 %1, %1, %5\n\t"
      "s_mov_b64 exec, -1\n\t"
      : "+v"(am0), "+v"(am1), "+v"(ax0), "+v"(ax1)
      : "v"(b0), "v"(b1), "s"(wc0), "s"(wc1), "s"(skc)
      : "scc");
}
// The same for one word of the self chunk / tail chunk, over the states that exist.
__device__ __forceinline__ void bwd_acc_word_masked(double& am, double& ax, double b, uint64_t wc, uint64_t skc,
                                                    uint32_t wi, const WordMasks& m)
{
  uint64_t tmp, vm;
  asm volatile(RELATE_VALID_TO_EXEC
               "s_mov_b64 %[vm], exec\n\t"
               "s_andn2_b64 %[tmp], %[skc], %[wc]\n\t"
               "s_and_b64 exec, %[tmp], %[vm]\n\t"
               "v_add_f64 %[ax], %[ax], %[b]\n\t"
               "s_orn2_b64 %[tmp], %[wc], %[skc]\n\t"
               "s_and_b64 exec, %[tmp], %[vm]\n\t"
               "v_add_f64 %[am], %[am], %[b]\n\t"
               "s_mov_b64 exec, -1\n\t"
               : [am] "+v"(am), [ax] "+v"(ax), [tmp] "=&s"(tmp), [vm] "=&s"(vm)
               : [b] "v"(b), [wc] "s"(wc), [skc] "s"(skc), [wi] "s"(wi), [kw] "s"(m.kw), [fw] "s"(m.fw), [uw] "s"(m.uw),
                 [ms] "s"(m.ms), [mt] "s"(m.mt)
               : "scc");
}

template <int MPL, int G>
__global__ void __launch_bounds__(64 * G) paint_kernel_x(
    const uint64_t* __restrict__ rows, const int32_t* __restrict__ dk_all, const double* __restrict__ coef_all,
    const double* __restrict__ nx_all, const int64_t* __restrict__ off, const int32_t* __restrict__ nd_all,
    const int32_t* __restrict__ bbj_all, const int32_t* __restrict__ bej_all, const int32_t* __restrict__ order,
    const int32_t* __restrict__ wb, uint8_t* __restrict__ out, const Scal p)
{
  static_assert(MPL % 8 == 0 && MPL <= 120, "states per lane come in pairs of 4-word chunks");
  constexpr int NCH = MPL / 4;  // 4-word chunks per wave and site (even, at most 30)
  __shared__ double red[2][G];
  __shared__ float stage[G * MPL * 64];  // staging of one stone payload
  int slot = 0;
  const int t = __builtin_amdgcn_readfirstlane(order[blockIdx.x]);
  const int k = p.k_begin + t;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wbase = wave * MPL;
  const int N = p.N, W = p.W, stride = p.row_stride;
  const int nd = __builtin_amdgcn_readfirstlane(nd_all[t]);
  const long long o = off[t];
  const int32_t* __restrict__ dk = dk_all + o;
  const double* __restrict__ coef = coef_all + o;
  const double* __restrict__ nx = nx_all + o;
  const int32_t* __restrict__ bbj = bbj_all + (long long)t * W;
  const int32_t* __restrict__ bej = bej_all + (long long)t * W;
  const size_t R = 56 + (size_t)8 * N;
  const int kw = k >> 6, kb = k & 63;
  const bool first_thread = threadIdx.x == 0;
  const PaintConsts& c = p.c;
  const int full_words = N >> 6;  // words whose 64 states all exist
  const int used_words = (N + 63) >> 6;
  // The words of this wave start here in every row; reads may run past the end of a row (into the
  // next one, or into the slack after the last): chunks without states are fetched but never used.
  const uint64_t* rows_w = rows + wbase;

  // Chunks (local index) whose partial sums need validity masks: the one holding the target itself and
  // the one where the states run out; chunks past that hold no state.  One bit per chunk says "not ordinary".
  const int l_self = (kw >> 2) - wave * NCH;
  const int l_tail = (full_words >> 2) - wave * NCH;  // first chunk with a word that is not full
  uint32_t special = 0;
  if (l_self >= 0 && l_self < NCH) special |= 1u << l_self;
  for (int cl = (l_tail < 0 ? 0 : l_tail); cl < NCH; cl++) special |= 1u << cl;
  special = sgpr32(special);
  WordMasks wm;
  wm.ms = sgpr64(valid_mask(kw, N, k));
  wm.mt = sgpr64(valid_mask(full_words, N, k));
  wm.kw = sgpr32((uint32_t)kw);
  wm.fw = sgpr32((uint32_t)full_words);
  wm.uw = sgpr32((uint32_t)used_words);
  const uint32_t wbase_s = sgpr32((uint32_t)wbase);

  // bit of the target itself at the two forced sites; it is '1' at every other site of its list
  const uint64_t sk_first = sgpr64(((rows[kw] >> kb) & 1ull) ? ~0ull : 0ull);
  const uint64_t sk_last = sgpr64(((rows[(size_t)(p.L - 1) * stride + kw] >> kb) & 1ull) ? ~0ull : 0ull);

  double a[MPL];
  u64x4 bufA, bufB;

  // Stone payload (src/fast_painting.cpp:368-388,:573-592): every lane drops its states as floats into
  // LDS (one ds_write per state, constant offsets), then the workgroup streams the N floats out with
  // coalesced stores; the target's own entry is 0 (:312,:507).  Keeping the copy loop rolled keeps
  // per-state addresses and lane masks out of the register file.
  auto store_states = [&](float* __restrict__ dst) __attribute__((always_inline)) {
    float* mine = stage + wbase * 64 + lane;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      mine[i * 64] = (float)a[i];
    });
    if constexpr (G > 1) __syncthreads();
    for (int n = threadIdx.x; n < N; n += 64 * G) dst[n] = (n == k) ? 0.0f : stage[n];
    if constexpr (G > 1) __syncthreads();
  };

  // ---------------- forward (src/fast_painting.cpp:214-408) ----------------
  {
    double acc = 0.0;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int wi = wbase + i;
      const uint64_t w = rows_w[i];
      const bool v = __builtin_amdgcn_inverse_ballot_w64(valid_mask(wi, N, k));
      const bool m = __builtin_amdgcn_inverse_ballot_w64(~w & sk_first);
      a[i] = m ? c.prior_mis : c.prior_match;
      acc += v ? a[i] : 0.0;
    });
    double S = block_sum<G>(acc, red, slot);
    double ls = 0.0;
    int wcur = 0;
    int stone_j = 0;  // list index at which the next alpha stone is due (window 0 starts at site 0)
    auto emit = [&](int j, int s) __attribute__((always_inline)) {
      while (stone_j == j) {
        uint8_t* rec = out + ((size_t)wcur * p.K + t) * R;
        if (first_thread) {
          put_u32(rec, (uint32_t)wb[wcur]);
          put_u32(rec + 4, (uint32_t)(wb[wcur + 1] - 1));
          put_block_header(rec + 8, N, s, (float)ls);
        }
        store_states(reinterpret_cast<float*>(rec + 32));
        wcur++;
        stone_j = wcur < W ? __builtin_amdgcn_readfirstlane(bbj[wcur]) : -1;
      }
    };
    emit(0, 0);
    double rx = coef[0] * S;
    double rho = c.rho_theta;
    pin_vgpr(rho);
    int s_cur = __builtin_amdgcn_readfirstlane(dk[1]);  // nd >= 2: sites 0 and L-1 are always listed
    const uint64_t* rp = rows_w + (size_t)s_cur * stride;
    sload4<0>(bufA, rp);
    swait(bufA);
    for (int j = 1; j < nd; j++) {
      const int s_next = __builtin_amdgcn_readfirstlane(dk[j + 1 < nd ? j + 1 : j]);
      const uint64_t* rp_next = rows_w + (size_t)s_next * stride;
      const uint64_t skm = edge_select(j, nd - 1, sk_last);
      uint32_t sp = special;
      opaque(sp);
      ls += nx[j - 1];
      double acc0 = 0.0, acc1 = 0.0, accs = 0.0;
      static_for<0, NCH>([&](auto ic) __attribute__((always_inline)) {
        constexpr int cl = decltype(ic)::value;
        u64x4& cur = (cl & 1) ? bufB : bufA;
        u64x4& nxt = (cl & 1) ? bufA : bufB;
        if constexpr (cl + 1 < NCH) sload4<(cl + 1) * 32>(nxt, rp);
        else sload4<0>(nxt, rp_next);
        fwd2_update(a[cl * 4 + 0], a[cl * 4 + 1], rx, rho, cur[0], cur[1], skm);
        fwd2_update(a[cl * 4 + 2], a[cl * 4 + 3], rx, rho, cur[2], cur[3], skm);
        if (__builtin_expect((sp >> cl) & 1u, 0)) {  // self / tail / empty chunk: sums over existing states only
          // (accumulators of their own: a value merged from both branches would be copied at every chunk)
          acc_word_masked(accs, a[cl * 4 + 0], wbase_s + cl * 4 + 0, wm);
          acc_word_masked(accs, a[cl * 4 + 1], wbase_s + cl * 4 + 1, wm);
          acc_word_masked(accs, a[cl * 4 + 2], wbase_s + cl * 4 + 2, wm);
          acc_word_masked(accs, a[cl * 4 + 3], wbase_s + cl * 4 + 3, wm);
        } else {
          acc0 += a[cl * 4 + 0];
          acc1 += a[cl * 4 + 1];
          acc0 += a[cl * 4 + 2];
          acc1 += a[cl * 4 + 3];
        }
        swait(nxt);
      });
      S = block_sum<G>((acc0 + acc1) + accs, red, slot);
      if (S < c.lo || S > c.hi) {
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) { a[decltype(ic)::value] /= S; });
        ls += log(S);
        S = 1.0;
      }
      rx = S * coef[j];
      emit(j, s_cur);
      s_cur = s_next;
      rp = rp_next;
    }
  }

  // ---------------- backward (src/fast_painting.cpp:410-599) ----------------
  {
    double ls = c.log_Nm1 - (double)nd * c.log_ntheta;
    const uint64_t* row_last = rows_w + (size_t)(p.L - 1) * stride;
    double acc = 0.0;
    uint32_t n_op = (uint32_t)N, k_op = (uint32_t)k;
    opaque(n_op);  // no sharing of per-word masks with the forward prologue (they would stay live in SGPRs)
    opaque(k_op);
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int wi = wbase + i;
      const uint64_t w = row_last[i];
      const bool v = __builtin_amdgcn_inverse_ballot_w64(valid_mask(wi, (int)n_op, (int)k_op));
      const bool m = __builtin_amdgcn_inverse_ballot_w64(~w & sk_last);
      a[i] = 1.0;
      acc += v ? (m ? c.theta_d : c.ntheta_d) : 0.0;
    });
    double B = block_sum<G>(acc, red, slot);
    int wcur = W - 1;
    int stone_j = __builtin_amdgcn_readfirstlane(bej[W - 1]);  // == nd-1: the last window ends at the last SNP
    // stones taken at the last SNP hold beta = 1 for every state, the target included (:427-432,:447-462)
    while (stone_j == nd - 1) {
      uint8_t* rec = out + ((size_t)wcur * p.K + t) * R;
      if (first_thread) put_block_header(rec + 32 + (size_t)4 * N, N, p.L - 1, (float)ls);
      float* dst = reinterpret_cast<float*>(rec + 56 + (size_t)4 * N);
      for (int n = threadIdx.x; n < N; n += 64 * G) dst[n] = 1.0f;
      wcur--;
      stone_j = wcur >= 0 ? __builtin_amdgcn_readfirstlane(bej[wcur]) : -1;
    }
    double rx = coef[nd - 1] * B;
    double rho = c.rho_theta;
    pin_vgpr(rho);
    u64x4 bufC, bufD;  // A/B: row handled one step earlier; C/D: current row
    int s_cur = __builtin_amdgcn_readfirstlane(dk[nd - 2]);
    const uint64_t* rp_n = row_last;
    const uint64_t* rp_c = rows_w + (size_t)s_cur * stride;
    sload4<0>(bufA, rp_n);
    sload4<0>(bufC, rp_c);
    swait(bufA);
    swait(bufC);
    for (int j = nd - 2; j >= 0; j--) {
      const int s_prev = __builtin_amdgcn_readfirstlane(dk[j > 0 ? j - 1 : 0]);
      const uint64_t* rp_p = rows_w + (size_t)s_prev * stride;
      const uint64_t skn = edge_select(j, nd - 2, sk_last);
      const uint64_t skc = edge_select(j, 0, sk_first);
      uint32_t sp = special;
      opaque(sp);
      ls += nx[j + 1];
      const double add_all = rx / c.ntheta_d;
      const double add_mis = rx / c.theta_d - add_all;
      double am0 = 0.0, am1 = 0.0, ax0 = 0.0, ax1 = 0.0, ams = 0.0, axs = 0.0;
      static_for<0, NCH>([&](auto ic) __attribute__((always_inline)) {
        constexpr int cl = decltype(ic)::value;
        u64x4& curn = (cl & 1) ? bufB : bufA;
        u64x4& nxtn = (cl & 1) ? bufA : bufB;
        u64x4& curc = (cl & 1) ? bufD : bufC;
        u64x4& nxtc = (cl & 1) ? bufC : bufD;
        if constexpr (cl + 1 < NCH) {
          sload4<(cl + 1) * 32>(nxtn, rp_n);
          sload4<(cl + 1) * 32>(nxtc, rp_c);
        } else {
          sload4<0>(nxtn, rp_c);
          sload4<0>(nxtc, rp_p);
        }
        bwd2_update(a[cl * 4 + 0], a[cl * 4 + 1], add_mis, add_all, rho, curn[0], curn[1], skn);
        bwd2_update(a[cl * 4 + 2], a[cl * 4 + 3], add_mis, add_all, rho, curn[2], curn[3], skn);
        if (__builtin_expect((sp >> cl) & 1u, 0)) {  // self / tail / empty chunk: sums over existing states only
          bwd_acc_word_masked(ams, axs, a[cl * 4 + 0], curc[0], skc, wbase_s + cl * 4 + 0, wm);
          bwd_acc_word_masked(ams, axs, a[cl * 4 + 1], curc[1], skc, wbase_s + cl * 4 + 1, wm);
          bwd_acc_word_masked(ams, axs, a[cl * 4 + 2], curc[2], skc, wbase_s + cl * 4 + 2, wm);
          bwd_acc_word_masked(ams, axs, a[cl * 4 + 3], curc[3], skc, wbase_s + cl * 4 + 3, wm);
        } else {
          bwd2_acc(am0, am1, ax0, ax1, a[cl * 4 + 0], a[cl * 4 + 1], curc[0], curc[1], skc);
          bwd2_acc(am0, am1, ax0, ax1, a[cl * 4 + 2], a[cl * 4 + 3], curc[2], curc[3], skc);
        }
        swait(nxtn);
        swait(nxtc);
      });
      // B = ntheta * (sum of beta' over matching states) + theta * (sum over mismatching states) (:509-517)
      B = block_sum<G>(c.ntheta_d * ((am0 + am1) + ams) + c.theta_d * ((ax0 + ax1) + axs), red, slot);
      if (B < c.lo || B > c.hi) {
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) { a[decltype(ic)::value] /= B; });
        ls += (double)fast_log_dev((float)B);  // float approximation on this side (:562)
        B = 1.0;
      }
      rx = B * coef[j];
      while (stone_j == j) {
        uint8_t* rec = out + ((size_t)wcur * p.K + t) * R;
        if (first_thread) put_block_header(rec + 32 + (size_t)4 * N, N, s_cur, (float)ls);
        store_states(reinterpret_cast<float*>(rec + 56 + (size_t)4 * N));
        wcur--;
        stone_j = wcur >= 0 ? __builtin_amdgcn_readfirstlane(bej[wcur]) : -1;
      }
      s_cur = s_prev;
      rp_n = rp_c;
      rp_c = rp_p;
    }
  }
}

template <int MPL, int G>
inline hipError_t launch_hand_t(const PaintArgs& a, hipStream_t st)
{
  Scal s;
  s.row_stride = a.row_stride;
  s.N = a.N;
  s.L = a.L;
  s.W = a.W;
  s.k_begin = a.k_begin;
  s.K = a.K;
  s.c = a.c;
  hipLaunchKernelGGL((paint_kernel_x<MPL, G>), dim3(a.K), dim3(64 * G), 0, st, a.rows, a.dk, a.coef, a.nx, a.off, a.nd,
                     a.bbj, a.bej, a.order, a.wb, a.out, s);
  return hipGetLastError();
}

// Dispatch on MPL for one G.
template <int G>
inline hipError_t launch_hand_g(const PaintArgs& a, int MPL, hipStream_t st)
{
  switch (MPL) {
    case 8: return launch_hand_t<8, G>(a, st);
    case 16: return launch_hand_t<16, G>(a, st);
    case 24: return launch_hand_t<24, G>(a, st);
    case 32: return launch_hand_t<32, G>(a, st);
    case 40: return launch_hand_t<40, G>(a, st);
    case 48: return launch_hand_t<48, G>(a, st);
    case 64: return launch_hand_t<64, G>(a, st);
    case 80:
      // the stone staging buffer of 8 waves x 80 states per lane would not fit in 160 KiB of LDS
      if constexpr (G * 80 * 256 + 2 * G * 8 <= 160 * 1024) return launch_hand_t<80, G>(a, st);
      else return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
}

}  // namespace hand
}  // namespace relate
