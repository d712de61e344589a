// Device helpers shared by the painting kernels (reductions, fast_log, record
// stores).  Internal header; included only by .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "paint_kernels.h"

namespace relate {
namespace dev {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ------------------------------------------------------------------ reductions
template <int CTRL>
__device__ __forceinline__ double dpp_pair_add(double x)
{
  int lo = __double2loint(x), hi = __double2hiint(x);
  int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return x + __hiloint2double(hi2, lo2);
}

__device__ __forceinline__ double read_lane(double x, int l)
{
  int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, same value (and same bits) in every lane: four
// exchange steps inside each 16-lane row (quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror, row_mirror), then the four row sums through SGPRs.
__device__ __forceinline__ double wave_sum(double x)
{
  x = dpp_pair_add<0xB1>(x);
  x = dpp_pair_add<0x4E>(x);
  x = dpp_pair_add<0x141>(x);
  x = dpp_pair_add<0x140>(x);
  const double r0 = read_lane(x, 0), r1 = read_lane(x, 16), r2 = read_lane(x, 32), r3 = read_lane(x, 48);
  return (r0 + r1) + (r2 + r3);
}

// Sum over the G waves of the workgroup.  `slot` alternates between two LDS
// rows from one call to the next, so one barrier per call is enough: a wave can
// only overwrite row p two calls later, after every wave has passed the barrier
// of the call in between, hence has finished reading row p.
template <int G>
__device__ __forceinline__ double block_sum(double x, double (*red)[G], int& slot)
{
  x = wave_sum(x);
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
    return s;
  }
}

__device__ __forceinline__ int wave_sum_int(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------ fast_log
// src/fast_log.hpp:6-21 with every float operation rounded on its own.
__device__ __forceinline__ float fast_log_dev(float val)
{
  int x = __float_as_int(val);
  const int log_2 = ((x >> 23) & 255) - 128;
  x &= ~(255 << 23);
  x += 127 << 23;
  val = __int_as_float(x);
  const float c1 = -1.0f / 3, c2 = 2.0f / 3;
  float t = __fmul_rn(c1, val);
  t = __fadd_rn(t, 2.0f);
  t = __fmul_rn(t, val);
  t = __fsub_rn(t, c2);
  t = __fadd_rn(t, (float)log_2);
  return __fmul_rn(t, 0.69314718f);
}

// ------------------------------------------------------------------ recursion helpers
struct Scal {
  int row_stride, N, L, W, k_begin, K;
  PaintConsts c;
};

// Lane mask of the states of word wi that take part in the recursion: n < N and n != k.
__device__ __forceinline__ uint64_t valid_mask(int wi, int N, int k)
{
  const int full = N >> 6, rem = N & 63;
  uint64_t m = (wi < full) ? ~0ull : ((wi == full && rem) ? ((1ull << rem) - 1ull) : 0ull);
  if (wi == (k >> 6)) m &= ~(1ull << (k & 63));
  return m;
}

__device__ __forceinline__ uint64_t row_word(const uint64_t* __restrict__ row, int wi, int row_stride)
{
  return (wi < row_stride) ? row[wi] : 0ull;
}

__device__ __forceinline__ void put_u32(uint8_t* p, uint32_t v) { *reinterpret_cast<uint32_t*>(p) = v; }

// Header of one block of a record (src/collapsed_matrix.hpp:240-252) by 32-bit stores
// (the beta block starts at 32+4N, which is only 4-byte aligned for odd N).
__device__ __forceinline__ void put_block_header(uint8_t* p, int N, int snp, float ls)
{
  put_u32(p, 1u);
  put_u32(p + 4, 0u);
  put_u32(p + 8, (uint32_t)N);
  put_u32(p + 12, 0u);
  put_u32(p + 16, (uint32_t)snp);
  put_u32(p + 20, __float_as_uint(ls));
}

template <int MPL>
__device__ __forceinline__ void store_payload(float* __restrict__ dst, const double (&a)[MPL], int wbase, int N)
{
  const int lane = lane_id();
#pragma unroll
  for (int i = 0; i < MPL; i++) {
    const int n = (wbase + i) * 64 + lane;
    if (n < N) dst[n] = (float)a[i];
  }
}

}  // namespace dev
}  // namespace relate
