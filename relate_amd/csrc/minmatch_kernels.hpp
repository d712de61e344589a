// The data-parallel part of MinMatch::Initialize (src/tree_builder.cpp:57-127; with sample ages :130-214): the two
// scans over the N x N distance matrix that open every tree build,
//   min_values[i] = min over l != i of d[i][l], plus threshold                         (:64-78)
//   every pair i < j with min_values[i] >= d[i][j] and min_values[j] >= d[j][i]          (:81-125)
// The second scan hands the host the pairs it would have found, compacted and in the order of the reference's
// double loop (i ascending, then j ascending), each with sym_dist = d[i][j] + d[j][i] (:95); the host then draws
// one random number per pair from its own generator, in that order, and keeps the candidates (:96-114) -- the part
// that is sequential by construction (rng.seed(1), :852).
//
// Plain HIP without matrix or packed instructions: the scans are bound by HBM (the matrix is read row-wise once
// per kernel; column entries d[j][i] are fetched only for the few j that pass the row test).
//
// The file also compiles for the host (RELATE_HOST_RUN: tests/hostrun/ runs these very kernels on CPU threads).
#pragma once
#ifndef RELATE_HOST_RUN
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace relate {
namespace minmatch {

constexpr int kScanThreads = 256;

// one wave per row
__global__ void row_min_kernel(const float* __restrict__ d, int N, float threshold, float* __restrict__ min_values)
{
  const int i = blockIdx.x;
  const int lane = threadIdx.x;
  const float* __restrict__ row = d + (long long)i * N;
  float mn = __builtin_inff();
  for (int l = lane; l < N; l += 64) {
    const float v = row[l];
    if (mn > v && l != i) mn = v;  // the reference's test, :70 (a NaN never becomes the minimum)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float other = __shfl_xor(mn, o, 64);
    if (mn > other) mn = other;
  }
  if (lane == 0) min_values[i] = mn + threshold;  // float + float, :75
}

// is (i, j), i < j, a candidate pair (:90-91)?  The column entry is read only when the row entry passes.
__device__ __forceinline__ bool is_candidate(const float* __restrict__ d, const float* __restrict__ min_values, int N, int i,
                                             int j, float mv_i, float* dij, float* dji)
{
  if (j >= N) return false;
  const float a = d[(long long)i * N + j];
  if (!(mv_i >= a)) return false;
  const float b = d[(long long)j * N + i];
  if (!(min_values[j] >= b)) return false;
  *dij = a;
  *dji = b;
  return true;
}

// one wave per row i: number of candidate pairs (i, j), j > i
__global__ void count_candidates_kernel(const float* __restrict__ d, const float* __restrict__ min_values, int N,
                                        int32_t* __restrict__ counts)
{
  const int i = blockIdx.x;
  const int lane = threadIdx.x;
  const float mv_i = min_values[i];
  int cnt = 0;
  for (int j0 = i + 1; j0 < N; j0 += 64) {
    float a, b;
    const bool c = is_candidate(d, min_values, N, i, j0 + lane, mv_i, &a, &b);
    cnt += __popcll(__ballot(c));
  }
  if (lane == 0) counts[i] = cnt;
}

// exclusive prefix sum of counts[0..N) into offsets[0..N], offsets[N] = total; one workgroup of kScanThreads
__global__ void scan_counts_kernel(const int32_t* __restrict__ counts, int N, int64_t* __restrict__ offsets)
{
  __shared__ long long part[kScanThreads];
  const int t = threadIdx.x;
  const int per = (N + kScanThreads - 1) / kScanThreads;
  const int lo = t * per < N ? t * per : N;
  const int hi = lo + per < N ? lo + per : N;
  long long s = 0;
  for (int k = lo; k < hi; k++) s += counts[k];
  part[t] = s;
  __syncthreads();
  // inclusive scan of the kScanThreads partial sums
  for (int o = 1; o < kScanThreads; o <<= 1) {
    const long long v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  long long run = part[t] - s;  // exclusive
  for (int k = lo; k < hi; k++) {
    offsets[k] = run;
    run += counts[k];
  }
  if (t == kScanThreads - 1) offsets[N] = part[t];
}

// one wave per row i: its candidate pairs in ascending j from offsets[i] on
__global__ void fill_candidates_kernel(const float* __restrict__ d, const float* __restrict__ min_values, int N,
                                       const int64_t* __restrict__ offsets, int64_t capacity, int32_t* __restrict__ cand_i,
                                       int32_t* __restrict__ cand_j, float* __restrict__ cand_dist)
{
  const int i = blockIdx.x;
  const int lane = threadIdx.x;
  const float mv_i = min_values[i];
  long long base = offsets[i];
  for (int j0 = i + 1; j0 < N; j0 += 64) {
    float a = 0.f, b = 0.f;
    const bool c = is_candidate(d, min_values, N, i, j0 + lane, mv_i, &a, &b);
    const unsigned long long m = __ballot(c);
    if (c) {
      const long long pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < capacity) {
        cand_i[pos] = i;
        cand_j[pos] = j0 + lane;
        cand_dist[pos] = a + b;  // float + float, :95
      }
    }
    base += __popcll(m);
  }
}

}  // namespace minmatch
}  // namespace relate
