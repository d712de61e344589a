// Instruction-cost microbenchmarks for gfx950 (development tool, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITERS 4000

template <int P>
__global__ void k(double* out, const uint64_t* masks, int iters, long long* cyc)
{
  const long long t_start = clock64();
  double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double rx = 1e-9, rho = 0.999999;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7, fr = 1e-9f;
  uint64_t m0 = masks[0], m1 = masks[1];
  uint32_t ml = __builtin_amdgcn_readfirstlane((uint32_t)m0), mh = __builtin_amdgcn_readfirstlane((uint32_t)(m0 >> 32));
  uint64_t sm0 = ((uint64_t)mh << 32) | ml;
  ml = __builtin_amdgcn_readfirstlane((uint32_t)m1); mh = __builtin_amdgcn_readfirstlane((uint32_t)(m1 >> 32));
  uint64_t sm1 = ((uint64_t)mh << 32) | ml;
  asm volatile("" : "+v"(rx), "+v"(rho), "+v"(fr));
  for (int it = 0; it < iters; it++) {
    if constexpr (P == 1) {  // 8 independent v_add_f64
      asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                   "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(rx));
    } else if constexpr (P == 2) {  // 8 independent v_fma_f64
      asm volatile("v_fma_f64 %0, %0, %9, %8\n v_fma_f64 %1, %1, %9, %8\n v_fma_f64 %2, %2, %9, %8\n v_fma_f64 %3, %3, %9, %8\n"
                   "v_fma_f64 %4, %4, %9, %8\n v_fma_f64 %5, %5, %9, %8\n v_fma_f64 %6, %6, %9, %8\n v_fma_f64 %7, %7, %9, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(rx), "v"(rho));
    } else if constexpr (P == 3) {  // 8 independent v_mul_f64
      asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                   "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(rho));
    } else if constexpr (P == 4) {  // 8 independent v_add_f32
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fr));
    } else if constexpr (P == 5) {  // forward pair block x4 (8 words): 24 f64 + 12 salu, accumulators included
      double c0 = 0, c1 = 0;
#define FWD(x, y)                                                                                     \
      asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n s_andn2_b64 exec, %8, %6\n v_mul_f64 %0, %0, %5\n"  \
                   "s_andn2_b64 exec, %8, %7\n v_mul_f64 %1, %1, %5\n s_mov_b64 exec, -1\n v_add_f64 %2, %2, %0\n v_add_f64 %3, %3, %1\n" \
                   : "+v"(x), "+v"(y), "+v"(c0), "+v"(c1) : "v"(rx), "v"(rho), "s"(sm0), "s"(sm1), "s"(sm1) : "scc")
      FWD(a0, a1); FWD(a2, a3); FWD(a4, a5); FWD(a6, a7);
      a0 += c0 * 1e-30; a1 += c1 * 1e-30;
    } else if constexpr (P == 6) {  // same arithmetic without exec toggles
      double c0 = 0, c1 = 0;
#define FWDN(x, y)                                                                                    \
      asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %0, %0, %5\n"             \
                   "v_mul_f64 %1, %1, %5\n v_add_f64 %2, %2, %0\n v_add_f64 %3, %3, %1\n"             \
                   : "+v"(x), "+v"(y), "+v"(c0), "+v"(c1) : "v"(rx), "v"(rho))
      FWDN(a0, a1); FWDN(a2, a3); FWDN(a4, a5); FWDN(a6, a7);
      a0 += c0 * 1e-30; a1 += c1 * 1e-30;
    } else if constexpr (P == 7) {  // 8 v_readlane_b32 + 4 exec writes
      uint32_t s0, s1, s2, s3, s4, s5, s6, s7;
      asm volatile("v_readlane_b32 %0, %8, 1\n v_readlane_b32 %1, %8, 2\n v_readlane_b32 %2, %8, 3\n v_readlane_b32 %3, %8, 4\n"
                   "v_readlane_b32 %4, %8, 5\n v_readlane_b32 %5, %8, 6\n v_readlane_b32 %6, %8, 7\n v_readlane_b32 %7, %8, 8\n"
                   : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(f0));
      asm volatile("" ::"s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7));
    } else if constexpr (P == 8) {  // 8 dependent v_add_f64 (latency)
      asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                   "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                   : "+v"(a0) : "v"(rx));
    } else if constexpr (P == 10) {  // 8 v_mov_b64
      asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n"
                   "v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(rx));
    } else if constexpr (P == 11) {  // backward batch of 4 words: 16 f64 + 14 salu
      double c0 = 0, c1 = 0, m0 = 0, m1 = 0;
      asm volatile(
          "s_andn2_b64 exec, %12, %10\n v_fma_f64 %0, %0, %9, %8\n s_andn2_b64 exec, %12, %11\n v_fma_f64 %1, %1, %9, %8\n"
          "s_andn2_b64 exec, %12, %10\n v_fma_f64 %2, %2, %9, %8\n s_andn2_b64 exec, %12, %11\n v_fma_f64 %3, %3, %9, %8\n"
          "s_orn2_b64 exec, %10, %12\n v_add_f64 %0, %0, %8\n s_orn2_b64 exec, %11, %12\n v_add_f64 %1, %1, %8\n"
          "s_orn2_b64 exec, %10, %12\n v_add_f64 %2, %2, %8\n s_orn2_b64 exec, %11, %12\n v_add_f64 %3, %3, %8\n"
          "s_mov_b64 exec, -1\n v_add_f64 %4, %4, %0\n v_add_f64 %5, %5, %1\n v_add_f64 %4, %4, %2\n v_add_f64 %5, %5, %3\n"
          "s_andn2_b64 exec, %12, %11\n v_add_f64 %6, %6, %0\n s_andn2_b64 exec, %12, %10\n v_add_f64 %7, %7, %1\n"
          "s_andn2_b64 exec, %12, %11\n v_add_f64 %6, %6, %2\n s_andn2_b64 exec, %12, %10\n v_add_f64 %7, %7, %3\n"
          "s_mov_b64 exec, -1\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(m0), "+v"(m1)
          : "v"(rx), "v"(rho), "s"(sm0), "s"(sm1), "s"(sm1) : "scc");
      a4 += (c0 + c1 + m0 + m1) * 1e-30;
    } else if constexpr (P == 12) {  // DPP reduction step x4: 8 dpp movs + 4 f64 adds, dependent, with the wait states
      int lo = __double2loint(a0), hi = __double2hiint(a0), tl, th;
      asm volatile(
          "s_nop 1\n v_mov_b32_dpp %2, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
          "s_nop 1\n v_mov_b32_dpp %2, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
          "s_nop 1\n v_mov_b32_dpp %2, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
          "s_nop 1\n v_mov_b32_dpp %2, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
          : "+v"(lo), "+v"(hi), "=&v"(tl), "=&v"(th));
      a0 = __hiloint2double(hi, lo);
    } else if constexpr (P == 13) {  // 8 SALU 64-bit logic ops on SGPRs (no exec)
      uint64_t t0, t1;
      asm volatile("s_andn2_b64 %0, %2, %3\n s_andn2_b64 %1, %3, %2\n s_andn2_b64 %0, %2, %3\n s_andn2_b64 %1, %3, %2\n"
                   "s_andn2_b64 %0, %2, %3\n s_andn2_b64 %1, %3, %2\n s_andn2_b64 %0, %2, %3\n s_andn2_b64 %1, %3, %2\n"
                   : "=&s"(t0), "=&s"(t1) : "s"(sm0), "s"(sm1) : "scc");
      asm volatile("" ::"s"(t0), "s"(t1));
    } else if constexpr (P == 14) {  // 8 exec writes
      asm volatile("s_andn2_b64 exec, %0, %1\n s_andn2_b64 exec, %1, %0\n s_andn2_b64 exec, %0, %1\n s_andn2_b64 exec, %1, %0\n"
                   "s_andn2_b64 exec, %0, %1\n s_andn2_b64 exec, %1, %0\n s_andn2_b64 exec, %0, %1\n s_mov_b64 exec, -1\n"
                   :: "s"(sm0), "s"(sm1) : "scc");
    } else if constexpr (P == 15) {  // 8 SALU 32-bit ops
      uint32_t t0, t1;
      asm volatile("s_add_u32 %0, %2, %3\n s_add_u32 %1, %3, %2\n s_add_u32 %0, %2, %3\n s_add_u32 %1, %3, %2\n"
                   "s_add_u32 %0, %2, %3\n s_add_u32 %1, %3, %2\n s_add_u32 %0, %2, %3\n s_add_u32 %1, %3, %2\n"
                   : "=&s"(t0), "=&s"(t1) : "s"((uint32_t)sm0), "s"((uint32_t)sm1) : "scc");
      asm volatile("" ::"s"(t0), "s"(t1));
    } else if constexpr (P == 16) {  // backward batch, fewer exec writes: unmasked add, masked fma (16 f64 + 10 s)
      double c0 = 0, c1 = 0, m0 = 0, m1 = 0;
      asm volatile(
          "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
          "s_andn2_b64 exec, %12, %10\n v_fma_f64 %0, %0, %9, %8\n s_andn2_b64 exec, %12, %11\n v_fma_f64 %1, %1, %9, %8\n"
          "s_andn2_b64 exec, %12, %10\n v_fma_f64 %2, %2, %9, %8\n s_andn2_b64 exec, %12, %11\n v_fma_f64 %3, %3, %9, %8\n"
          "s_mov_b64 exec, -1\n v_add_f64 %4, %4, %0\n v_add_f64 %5, %5, %1\n v_add_f64 %4, %4, %2\n v_add_f64 %5, %5, %3\n"
          "s_andn2_b64 exec, %12, %11\n v_add_f64 %6, %6, %0\n s_andn2_b64 exec, %12, %10\n v_add_f64 %7, %7, %1\n"
          "s_andn2_b64 exec, %12, %11\n v_add_f64 %6, %6, %2\n s_andn2_b64 exec, %12, %10\n v_add_f64 %7, %7, %3\n"
          "s_mov_b64 exec, -1\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(m0), "+v"(m1)
          : "v"(rx), "v"(rho), "s"(sm0), "s"(sm1), "s"(sm1) : "scc");
      a4 += (c0 + c1 + m0 + m1) * 1e-30;
    } else if constexpr (P == 17) {  // backward mix, no masks (16 f64)
      double c0 = 0, c1 = 0, m0 = 0, m1 = 0;
      asm volatile(
          "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
          "v_fma_f64 %0, %0, %9, %8\n v_fma_f64 %1, %1, %9, %8\n"
          "v_fma_f64 %2, %2, %9, %8\n v_fma_f64 %3, %3, %9, %8\n"
          "v_add_f64 %4, %4, %0\n v_add_f64 %5, %5, %1\n v_add_f64 %4, %4, %2\n v_add_f64 %5, %5, %3\n"
          "v_add_f64 %6, %6, %0\n v_add_f64 %7, %7, %1\n"
          "v_add_f64 %6, %6, %2\n v_add_f64 %7, %7, %3\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(m0), "+v"(m1)
          : "v"(rx), "v"(rho), "s"(sm0), "s"(sm1), "s"(sm1) : "scc");
      a4 += (c0 + c1 + m0 + m1) * 1e-30;
    } else if constexpr (P == 18) {  // backward mix, masks on fma only (16 f64 + 5 s)
      double c0 = 0, c1 = 0, m0 = 0, m1 = 0;
      asm volatile(
          "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
          "s_andn2_b64 exec, %12, %10\n v_fma_f64 %0, %0, %9, %8\n s_andn2_b64 exec, %12, %11\n v_fma_f64 %1, %1, %9, %8\n"
          "s_andn2_b64 exec, %12, %10\n v_fma_f64 %2, %2, %9, %8\n s_andn2_b64 exec, %12, %11\n v_fma_f64 %3, %3, %9, %8\n"
          "s_mov_b64 exec, -1\n"
          "v_add_f64 %4, %4, %0\n v_add_f64 %5, %5, %1\n v_add_f64 %4, %4, %2\n v_add_f64 %5, %5, %3\n"
          "v_add_f64 %6, %6, %0\n v_add_f64 %7, %7, %1\n"
          "v_add_f64 %6, %6, %2\n v_add_f64 %7, %7, %3\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(m0), "+v"(m1)
          : "v"(rx), "v"(rho), "s"(sm0), "s"(sm1), "s"(sm1) : "scc");
      a4 += (c0 + c1 + m0 + m1) * 1e-30;
    } else if constexpr (P == 19) {  // backward mix, masks on masked sums only (16 f64 + 5 s)
      double c0 = 0, c1 = 0, m0 = 0, m1 = 0;
      asm volatile(
          "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
          "v_fma_f64 %0, %0, %9, %8\n v_fma_f64 %1, %1, %9, %8\n"
          "v_fma_f64 %2, %2, %9, %8\n v_fma_f64 %3, %3, %9, %8\n"
          "v_add_f64 %4, %4, %0\n v_add_f64 %5, %5, %1\n v_add_f64 %4, %4, %2\n v_add_f64 %5, %5, %3\n"
          "s_andn2_b64 exec, %12, %11\n v_add_f64 %6, %6, %0\n s_andn2_b64 exec, %12, %10\n v_add_f64 %7, %7, %1\n"
          "s_andn2_b64 exec, %12, %11\n v_add_f64 %6, %6, %2\n s_andn2_b64 exec, %12, %10\n v_add_f64 %7, %7, %3\n"
          "s_mov_b64 exec, -1\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(m0), "+v"(m1)
          : "v"(rx), "v"(rho), "s"(sm0), "s"(sm1), "s"(sm1) : "scc");
      a4 += (c0 + c1 + m0 + m1) * 1e-30;
    } else if constexpr (P == 20) {  // 8 independent FP64 MFMAs (4 blocks of 4x4x4)
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %8, %9, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %8, %9, %1\n v_mfma_f64_4x4x4_4b_f64 %2, %8, %9, %2\n"
                   "v_mfma_f64_4x4x4_4b_f64 %3, %8, %9, %3\n v_mfma_f64_4x4x4_4b_f64 %4, %8, %9, %4\n v_mfma_f64_4x4x4_4b_f64 %5, %8, %9, %5\n"
                   "v_mfma_f64_4x4x4_4b_f64 %6, %8, %9, %6\n v_mfma_f64_4x4x4_4b_f64 %7, %8, %9, %7\n s_nop 7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(rx), "v"(rho));
    } else if constexpr (P == 21) {  // 1 MFMA + 7 independent v_add_f64
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %8, %9, %0\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                   "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(rx), "v"(rho));
    } else if constexpr (P == 22) {  // 2 MFMA + 6 independent v_add_f64
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %8, %9, %0\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                   "v_mfma_f64_4x4x4_4b_f64 %4, %8, %9, %4\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(rx), "v"(rho));
    } else if constexpr (P == 23) {  // a chain of 3 dependent MFMAs (the reduction) + 7 independent v_add_f64 spread between them
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %0, %9, 0\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n s_nop 3\n"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %0, %9, 0\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n s_nop 3\n"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %0, %9, 0\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n s_nop 2\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(rx), "v"(rho));
    } else if constexpr (P == 24) {  // the same chain alone
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %0, %2, 0\n s_nop 5\n"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %0, %2, 0\n s_nop 5\n"
                   "v_mfma_f64_4x4x4_4b_f64 %0, %0, %2, 0\n s_nop 5\n"
                   : "+v"(a0) : "v"(rx), "v"(rho));
    } else if constexpr (P == 9) {  // 8 v_cndmask_b32 with SGPR mask
      asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                   "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fr), "s"(sm0));
    }
  }
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = clock64() - t_start;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

// Which input lanes reach which output lane of one 4-block 4x4x4 FP64 MFMA when the other operand is all ones:
// reach[(side*64 + p)*64 + lane] = output of lane `lane` for the one-hot input at lane p (side 0: data as A, 1: as B).
__global__ void mfma_layout(double* reach)
{
  const int lane = threadIdx.x;
  for (int side = 0; side < 2; side++)
    for (int p = 0; p < 64; p++) {
      const double hot = lane == p ? 1.0 : 0.0;
      const double d = side == 0 ? __builtin_amdgcn_mfma_f64_4x4x4f64(hot, 1.0, 0.0, 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, hot, 0.0, 0, 0, 0);
      reach[(side * 64 + p) * 64 + lane] = d;
    }
}

static void layout_report()
{
  double* d; hipMalloc(&d, 2 * 64 * 64 * 8);
  hipLaunchKernelGGL(mfma_layout, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(2 * 64 * 64);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  for (int side = 0; side < 2; side++) {
    printf("mfma_f64_4x4x4 layout, data as %s, other operand ones: output lane <- input lanes\n", side ? "B" : "A");
    for (int lane = 0; lane < 64; lane++) {
      printf("  out %2d <-", lane);
      for (int p = 0; p < 64; p++) if (h[(side * 64 + p) * 64 + lane] != 0.0) printf(" %d", p);
      printf("\n");
    }
    // three applications: count of each input lane in every output lane
    std::vector<double> m(64 * 64), t(64 * 64), u(64 * 64);
    for (int o = 0; o < 64; o++) for (int p = 0; p < 64; p++) m[o * 64 + p] = h[(side * 64 + p) * 64 + o];
    auto mul = [](const std::vector<double>& x, const std::vector<double>& y, std::vector<double>& z) {
      for (int i = 0; i < 64; i++) for (int j = 0; j < 64; j++) { double s = 0; for (int q = 0; q < 64; q++) s += x[i * 64 + q] * y[q * 64 + j]; z[i * 64 + j] = s; } };
    mul(m, m, t); mul(m, t, u);
    for (int o : {0, 1, 4, 16, 63}) { printf("  after 3 applications, out %2d counts:", o); for (int p = 0; p < 64; p++) printf(" %g", u[o * 64 + p]); printf("\n"); }
  }
  hipFree(d);
}

template <int P>
void run(const char* name, int instrs, double* d_out, uint64_t* d_masks)
{
  static long long* d_cyc = nullptr;
  if (!d_cyc) hipMalloc(&d_cyc, sizeof(long long) * 512 * 16);
  for (int waves : {1, 2, 4, 8}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // one block per CU up to 4 waves per SIMD; two blocks of 1024 threads per CU for 8
    dim3 grid(waves == 8 ? 512 : 256), block(waves == 8 ? 1024 : 256 * waves);
    hipLaunchKernelGGL(k<P>, grid, block, 0, 0, d_out, d_masks, 100, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, grid, block, 0, 0, d_out, d_masks, ITERS, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nw = grid.x * block.x / 64;
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), d_cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h) mean += (double)v;
    mean /= nw;
    double cyc = ms * 1e-3 * 2.4e9 / ((double)ITERS * instrs * waves);
    printf("%-28s waves/SIMD %d  %8.3f ms  %6.2f cycles (at 2.4 GHz) per instr per SIMD; clock64 ticks per instr per SIMD %6.2f (ticks/ms %.0f)\n",
           name, waves, ms, cyc, mean / ((double)ITERS * instrs * waves), mean / ms);
  }
}

int main()
{
  double* d_out; uint64_t* d_masks;
  hipMalloc(&d_out, 512 * 1024 * 8);
  hipMalloc(&d_masks, 16);
  uint64_t h[2] = {0x5555aaaa3333ccccull, 0x0f0ff0f012345678ull};
  hipMemcpy(d_masks, h, 16, hipMemcpyHostToDevice);
  layout_report();
  run<20>("mfma_f64_4x4x4 x8 indep", 8, d_out, d_masks);
  run<21>("1 mfma + 7 v_add_f64", 8, d_out, d_masks);
  run<22>("2 mfma + 6 v_add_f64", 8, d_out, d_masks);
  run<23>("3-mfma chain + 7 v_add_f64", 10, d_out, d_masks);
  run<24>("3-mfma chain alone", 3, d_out, d_masks);
  run<1>("v_add_f64 x8 indep", 8, d_out, d_masks);
  run<2>("v_fma_f64 x8 indep", 8, d_out, d_masks);
  run<3>("v_mul_f64 x8 indep", 8, d_out, d_masks);
  run<4>("v_add_f32 x8 indep", 8, d_out, d_masks);
  run<8>("v_add_f64 x8 dependent", 8, d_out, d_masks);
  run<5>("fwd pair block (24 f64+12 s)", 36, d_out, d_masks);
  run<6>("same, no exec toggles (24)", 24, d_out, d_masks);
  run<7>("v_readlane_b32 x8", 8, d_out, d_masks);
  run<9>("v_cndmask_b32 sgpr mask x8", 8, d_out, d_masks);
  run<10>("v_mov_b64 x8", 8, d_out, d_masks);
  run<11>("bwd batch (16 f64+14 s)", 30, d_out, d_masks);
  run<12>("dpp reduce (8 dpp+8 f32)", 16, d_out, d_masks);
  run<13>("s_andn2_b64 sgpr x8", 8, d_out, d_masks);
  run<14>("exec writes x8", 8, d_out, d_masks);
  run<15>("s_add_u32 x8", 8, d_out, d_masks);
  run<16>("bwd batch B (16 f64+10 s)", 26, d_out, d_masks);
  run<17>("bwd mix no masks (16)", 16, d_out, d_masks);
  run<18>("bwd mix fma masks (16+5)", 21, d_out, d_masks);
  run<19>("bwd mix sum masks (16+5)", 21, d_out, d_masks);
  return 0;
}
