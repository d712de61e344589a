// Test-only probes of the inline-asm building blocks of the hand-scheduled recursion kernel
// (relate_amd/csrc/paint_hand.hpp): each runs one helper on one wave and writes what it produced.
#include "../../relate_amd/csrc/paint_hand.hpp"

using namespace relate::hand;

__global__ void probe_kernel(const uint64_t* __restrict__ words /*8 words*/, const double* __restrict__ in /*[4][64]*/,
                             double* __restrict__ out /*[16][64]*/, uint64_t* __restrict__ wout /*8*/, double rx,
                             double rho, double add_mis, double add_all, uint64_t skm_in, uint64_t skc_in,
                             uint32_t kw, uint32_t fw, uint32_t uw, uint64_t ms_in, uint64_t mt_in)
{
  const int lane = threadIdx.x;
  u64x4 bufA, bufB;
  sload4<0>(bufA, words);
  sload4<32>(bufB, words);
  swait(bufA);
  swait(bufB);
  {
    // separate loads for the echo of the words: a buffer that also feeds a vector store may be kept in
    // VGPRs by the compiler, and could then not be an "s" operand
    u64x4 bufC, bufD;
    sload4<0>(bufC, words);
    sload4<32>(bufD, words);
    swait(bufC);
    swait(bufD);
    if (lane == 0) {
      wout[0] = bufC[0]; wout[1] = bufC[1]; wout[2] = bufC[2]; wout[3] = bufC[3];
      wout[4] = bufD[0]; wout[5] = bufD[1]; wout[6] = bufD[2]; wout[7] = bufD[3];
    }
  }
  const uint64_t skm = sgpr64(skm_in), skc = sgpr64(skc_in);
  pin_vgpr(rho);
  // forward update + plain accumulate
  double a0 = in[0 * 64 + lane], a1 = in[1 * 64 + lane], a2 = in[2 * 64 + lane], a3 = in[3 * 64 + lane];
  fwd2_update(a0, a1, rx, rho, bufA[0], bufA[1], skm);
  fwd2_update(a2, a3, rx, rho, bufA[2], bufA[3], skm);
  out[0 * 64 + lane] = a0; out[1 * 64 + lane] = a1; out[2 * 64 + lane] = a2; out[3 * 64 + lane] = a3;
  // masked accumulate of words 0..3
  WordMasks wm;
  wm.ms = sgpr64(ms_in); wm.mt = sgpr64(mt_in); wm.kw = sgpr32(kw); wm.fw = sgpr32(fw); wm.uw = sgpr32(uw);
  const uint32_t zero = sgpr32(0);
  double acc = 0.0;
  acc_word_masked(acc, a0, zero + 0, wm);
  acc_word_masked(acc, a1, zero + 1, wm);
  acc_word_masked(acc, a2, zero + 2, wm);
  acc_word_masked(acc, a3, zero + 3, wm);
  out[4 * 64 + lane] = acc;
  // backward update (wn = bufA) and accumulate (wc = bufB)
  double b0 = in[0 * 64 + lane], b1 = in[1 * 64 + lane], b2 = in[2 * 64 + lane], b3 = in[3 * 64 + lane];
  bwd2_update(b0, b1, add_mis, add_all, rho, bufA[0], bufA[1], skm);
  bwd2_update(b2, b3, add_mis, add_all, rho, bufA[2], bufA[3], skm);
  out[5 * 64 + lane] = b0; out[6 * 64 + lane] = b1; out[7 * 64 + lane] = b2; out[8 * 64 + lane] = b3;
  double am0 = 0, am1 = 0, ax0 = 0, ax1 = 0;
  bwd2_acc(am0, am1, ax0, ax1, b0, b1, bufB[0], bufB[1], skc);
  bwd2_acc(am0, am1, ax0, ax1, b2, b3, bufB[2], bufB[3], skc);
  out[9 * 64 + lane] = am0; out[10 * 64 + lane] = am1; out[11 * 64 + lane] = ax0; out[12 * 64 + lane] = ax1;
  double ams = 0, axs = 0;
  bwd_acc_word_masked(ams, axs, b0, bufB[0], skc, zero + 0, wm);
  bwd_acc_word_masked(ams, axs, b1, bufB[1], skc, zero + 1, wm);
  bwd_acc_word_masked(ams, axs, b2, bufB[2], skc, zero + 2, wm);
  bwd_acc_word_masked(ams, axs, b3, bufB[3], skc, zero + 3, wm);
  out[13 * 64 + lane] = ams; out[14 * 64 + lane] = axs;
  out[15 * 64 + lane] = relate::dev::wave_sum(in[lane]);
}

extern "C" int probe_run(const uint64_t* words, const double* in, double* out, uint64_t* wout, double rx, double rho,
                         double add_mis, double add_all, uint64_t skm, uint64_t skc, uint32_t kw, uint32_t fw,
                         uint32_t uw, uint64_t ms, uint64_t mt)
{
  uint64_t *d_words, *d_wout;
  double *d_in, *d_out;
  if (hipMalloc(&d_words, 16 * 8) || hipMalloc(&d_wout, 8 * 8) || hipMalloc(&d_in, 4 * 64 * 8) || hipMalloc(&d_out, 16 * 64 * 8))
    return 1;
  hipMemcpy(d_words, words, 8 * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_in, in, 4 * 64 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, d_words, d_in, d_out, d_wout, rx, rho, add_mis, add_all, skm,
                     skc, kw, fw, uw, ms, mt);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  hipMemcpy(out, d_out, 16 * 64 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(wout, d_wout, 8 * 8, hipMemcpyDeviceToHost);
  hipFree(d_words); hipFree(d_wout); hipFree(d_in); hipFree(d_out);
  return 0;
}
