// Lane maps and summation order of v_mfma_f64_4x4x4_4b_f64 used as a cross-lane adder (one operand all ones):
// prints, for each output lane, which input lanes were summed (exact integer inputs), for the four ways of
// chaining two of them, and checks the order of the four additions against sequential fused multiply-adds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>

__global__ void k_lanes(const double* x, double* out)
{
  const int l = threadIdx.x;
  const double v = x[l], one = 1.0;
  double d_a = __builtin_amdgcn_mfma_f64_4x4x4f64(v, one, 0.0, 0, 0, 0);    // data as A
  double d_b = __builtin_amdgcn_mfma_f64_4x4x4f64(one, v, 0.0, 0, 0, 0);    // data as B
  double d_aa = __builtin_amdgcn_mfma_f64_4x4x4f64(d_a, one, 0.0, 0, 0, 0);
  double d_ab = __builtin_amdgcn_mfma_f64_4x4x4f64(one, d_a, 0.0, 0, 0, 0);
  double d_ba = __builtin_amdgcn_mfma_f64_4x4x4f64(d_b, one, 0.0, 0, 0, 0);
  double d_bb = __builtin_amdgcn_mfma_f64_4x4x4f64(one, d_b, 0.0, 0, 0, 0);
  out[0 * 64 + l] = d_a;
  out[1 * 64 + l] = d_b;
  out[2 * 64 + l] = d_aa;
  out[3 * 64 + l] = d_ab;
  out[4 * 64 + l] = d_ba;
  out[5 * 64 + l] = d_bb;
}


// ---- timing: a 20-word sweep stand-in (60 FP64 instructions) + a 64-lane sum, five waves per SIMD ----
#define DPP_STEP(ctrl)                                                     \
  "s_nop 1\n\t"                                                            \
  "v_mov_b32_dpp v102, v100 " ctrl " row_mask:0xf bank_mask:0xf\n\t"       \
  "v_mov_b32_dpp v103, v101 " ctrl " row_mask:0xf bank_mask:0xf\n\t"     \
  "v_add_f64 v[100:101], v[100:101], v[102:103]\n\t"
#define DPP_ROWS                                                           \
  "s_nop 1\n\t"                                                            \
  "v_mov_b32_dpp v102, v100 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"   \
  "v_mov_b32_dpp v103, v101 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
  "s_mov_b32 exec_lo, 0xffff0000\n\t"                                      \
  "s_mov_b32 exec_hi, 0xffff0000\n\t"                                      \
  "v_add_f64 v[100:101], v[100:101], v[102:103]\n\t"                                         \
  "s_mov_b64 exec, -1\n\t"                                                 \
  "s_nop 0\n\t"                                                            \
  "v_mov_b32_dpp v102, v100 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"   \
  "v_mov_b32_dpp v103, v101 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
  "s_mov_b32 exec_lo, 0\n\t"                                               \
  "v_add_f64 v[100:101], v[100:101], v[102:103]\n\t"                                         \
  "s_mov_b64 exec, -1\n\t"                                                 \
  "s_nop 0\n\t"                                                            \
  "v_readlane_b32 vcc_lo, v100, 63\n\t"                                    \
  "v_readlane_b32 vcc_hi, v101, 63\n\t"                                   \
  "s_nop 1\n\t"                                                            \
  "v_mov_b64 v[100:101], vcc\n\t"

template <int MODE>
__global__ void __launch_bounds__(256) k_time(double* out, int iters, double c0, double d0)
{
  double a[20];
  for (int w = 0; w < 20; w++) a[w] = 1.0 + 1e-3 * (threadIdx.x + w);
  double c = c0, d = d0;
  const double one = 1.0;
  for (int it = 0; it < iters; it++) {
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < 20; w++) {
      a[w] = __dadd_rn(__dmul_rn(a[w], c), d);
      a[w] = __dmul_rn(a[w], d0 + 0.9995);
      sum = __dadd_rn(sum, a[w]);
    }
    double x;
    asm volatile("v_mov_b64 v[100:101], %0" ::"v"(sum) : "v100", "v101");
    if constexpr (MODE == 0) {
      asm volatile(DPP_STEP("quad_perm:[1,0,3,2]") DPP_STEP("quad_perm:[2,3,0,1]") DPP_STEP("row_half_mirror")
                       DPP_STEP("row_mirror") DPP_ROWS
                   ::: "vcc", "v100", "v101", "v102", "v103");
    } else {
      asm volatile("s_nop 1\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 v[102:103], v[100:101], %[one], 0\n\t"
                   "s_nop 5\n\t"
                   "v_mfma_f64_4x4x4_4b_f64 v[100:101], %[one], v[102:103], 0\n\t"
                   "s_nop 5\n\t" DPP_ROWS
                   :: [one] "v"(one)
                   : "vcc", "v100", "v101", "v102", "v103");
    }
    asm volatile("v_mov_b64 %0, v[100:101]" : "=v"(x)::"v100", "v101");
    c = __dadd_rn(__dmul_rn(x, 1e-30), c0);
  }
  double r = c;
  for (int w = 0; w < 20; w++) r += a[w];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

static void run_timing()
{
  const int blocks = 256 * 5, iters = 20000;
  double* o;
  hipMalloc(&o, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (mode == 0) k_time<0><<<blocks, 256>>>(o, iters, 0.999, 1e-3);
      else k_time<1><<<blocks, 256>>>(o, iters, 0.999, 1e-3);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double h[4];
      hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
      printf("timing %s: %.2f ms for %d sites x 5 waves per SIMD (%.1f ns per site-round)  out %.17g\n", mode ? "2 MFMA + 2 row steps" : "6 DPP steps", ms,
             iters, ms * 1e6 / iters, h[0]);
    }
}

int main()
{
  double *dx, *dout;
  hipMalloc(&dx, 64 * 8);
  hipMalloc(&dout, 6 * 64 * 8);
  double hx[64], ho[6 * 64];
  const char* names[6] = {"A=x,B=1", "A=1,B=x", "A=(A=x),B=1", "A=1,B=(A=x)", "A=(B=x),B=1", "A=1,B=(B=x)"};
  // pass 0: lanes 0..31 carry bit l, pass 1: lanes 32..63 carry bit l-32
  uint64_t mask[6][64];
  memset(mask, 0, sizeof mask);
  for (int p = 0; p < 2; p++) {
    for (int l = 0; l < 64; l++) hx[l] = (l / 32 == p) ? (double)(1u << (l & 31)) : 0.0;
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    k_lanes<<<1, 64>>>(dx, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    for (int v = 0; v < 6; v++)
      for (int l = 0; l < 64; l++) {
        // second-level results may count a lane several times: decode by digits only for first-level
        double r = ho[v * 64 + l];
        if (v < 2) mask[v][l] |= (uint64_t)(uint32_t)r << (32 * p);
      }
  }
  for (int v = 0; v < 2; v++) {
    printf("%s\n", names[v]);
    for (int l = 0; l < 64; l++) {
      printf("  lane %2d <-", l);
      for (int s = 0; s < 64; s++)
        if (mask[v][l] >> s & 1) printf(" %d", s);
      printf("\n");
    }
  }
  // second level: x = 1 in one row of 16 lanes at a time tells the multiplicity; use distinct small primes' ... simply
  // weights w_l = 1 + l / 1024 are not exact; use x_l = 2^(l mod 16) in row r only (4 passes) and print sums.
  for (int r = 0; r < 4; r++) {
    for (int l = 0; l < 64; l++) hx[l] = (l / 16 == r) ? (double)(1u << (l & 15)) : 0.0;
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    k_lanes<<<1, 64>>>(dx, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    for (int v = 2; v < 6; v++) {
      printf("%s, input row %d (65535 = each lane of the row once):", names[v], r);
      for (int l = 0; l < 64; l++) printf(" %g", ho[v * 64 + l]);
      printf("\n");
    }
  }
  // order of the four additions: inputs with very different magnitudes in the four summed lanes
  {
    srand(5);
    for (int l = 0; l < 64; l++) hx[l] = ldexp((double)rand() / RAND_MAX + 0.5, (rand() % 7) - 3) * ((rand() & 1) ? 1 : -1);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    k_lanes<<<1, 64>>>(dx, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    for (int v = 0; v < 2; v++) {
      int ok_seq = 0, ok_rev = 0, ok_pair = 0;
      for (int l = 0; l < 64; l++) {
        // the summed lanes from the masks, ascending
        int s[4], n = 0;
        for (int q = 0; q < 64; q++)
          if (mask[v][l] >> q & 1) s[n < 4 ? n++ : 3] = q;
        volatile double a = 0.0;
        a = a + hx[s[0]]; a = a + hx[s[1]]; a = a + hx[s[2]]; a = a + hx[s[3]];
        volatile double b = 0.0;
        b = b + hx[s[3]]; b = b + hx[s[2]]; b = b + hx[s[1]]; b = b + hx[s[0]];
        volatile double c1 = hx[s[0]] + hx[s[1]], c2 = hx[s[2]] + hx[s[3]];
        volatile double c = c1 + c2;
        ok_seq += ho[v * 64 + l] == a;
        ok_rev += ho[v * 64 + l] == b;
        ok_pair += ho[v * 64 + l] == c;
      }
      printf("%s: lanes equal to ascending sequential sum %d, descending %d, pairwise %d of 64\n", names[v], ok_seq, ok_rev, ok_pair);
    }
  }
  run_timing();
  return 0;
}
