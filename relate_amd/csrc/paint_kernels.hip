// gfx950 (MI355X) kernels of the stepping-stone painting path.
//
// Layout in HBM
//   rows  [L][row_stride] u64 : haplotype matrix, one bit per state; word w of row s
//                               holds states 64w..64w+63, so a word IS the lane mask of
//                               the 64 states one wave-instruction updates.
//   cols  [N][col_stride] u64 : the transpose (bit s&63 of word s>>6 of column k), used
//                               only to enumerate a target's derived sites.
//   dk/coef/nx (flat, per target): derived-site list and the two per-gap terms.
//   out   [W][K][56+8N] bytes : records exactly as they lie in relate_<w>.bin.
//
// Recursion kernel: one workgroup of G waves owns one target; lane l of wave g
// keeps states 64(g*MPL+i)+l, i < MPL, as doubles in VGPRs for the whole pass.
// The per-site normaliser is a DPP row reduction + 4 readlanes (+ one LDS
// exchange when G > 1).  No MFMA: the recurrence is a rank-1 update plus a
// diagonal scaling per site, not a contraction.
//
// Compiled with -ffp-contract=off: every product and sum rounds on its own,
// as in the reference's scalar code; fused operations appear only where written.
#include "minmatch_kernels.hpp"
#include "paint_device.hpp"
#include "paint_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace relate {

namespace {

using namespace dev;

__global__ void fast_log_kernel(const float* __restrict__ in, float* __restrict__ out, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fast_log_dev(in[i]);
}

// ------------------------------------------------------------------ distance matrix
// DistanceMeasure::GetMatrix (src/anc_builder.cpp:113-211).  One workgroup per row n of the N x N matrix:
//   direct rows (n carries the derived allele at the SNP, or the SNP is the first / last of the chunk):
//     d = (fast_log(top[m]) + logscale) * scale                                   (:122-141)
//   other rows: the two topology rows around the SNP are mixed in probability space with the weights
//     of the genetic distances, rescaled to the larger of the two logscales        (:143-193)
//   then d[n][n] = 0 and the row minimum -- taken over all m, the diagonal's own value included, as
//   the reference does -- is subtracted from the off-diagonal entries              (:194-196).
// The arithmetic keeps the reference's types: weights in double, the mix in double, narrowed to float
// for fast_log, float from there on; scale = -1.0f (src/anc_builder.hpp:46).  The two exponentials of a
// mixed row (:171-172) are NOT taken here: the host computes them with the same libm call as the
// reference (exp on a double, narrowed) and passes them per row, so that every operation of an entry is
// an individually rounded IEEE operation and the matrix is the reference's bit for bit.
__device__ __forceinline__ float distance_entry(bool interp, bool prev_le_next, float tp, float tn, double wl, double wr,
                                                float e_pn, float e_np, float ls_prev, float ls_next)
{
  const float scale = -1.0f;
  if (!interp) return __fmul_rn(__fadd_rn(fast_log_dev(tp), ls_prev), scale);
  if (prev_le_next) {
    const double x = __dadd_rn(__dmul_rn(__dmul_rn(wl, (double)tp), (double)e_pn), __dmul_rn(wr, (double)tn));
    return __fmul_rn(__fadd_rn(fast_log_dev((float)x), ls_next), scale);
  }
  const double x = __dadd_rn(__dmul_rn(wl, (double)tp), __dmul_rn(__dmul_rn(wr, (double)tn), (double)e_np));
  return __fmul_rn(__fadd_rn(fast_log_dev((float)x), ls_prev), scale);
}

// The two logscales every row of the matrix needs, gathered for the host (which takes their exponentials):
// out[2n] = ls[row[n]], out[2n+1] = ls[row[n] + 1] where the row is mixed, else the same value again.
__global__ void gather_logscales_kernel(const float* __restrict__ ls, const int64_t* __restrict__ row,
                                        const uint8_t* __restrict__ interp_flag, int N, float* __restrict__ out)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const long long r0 = row[n];
  out[2 * n] = ls[r0];
  out[2 * n + 1] = ls[r0 + (interp_flag[n] ? 1 : 0)];
}

// ITEMS > 0: thread t keeps its entries m = t + 256 i, i < ITEMS (N <= 256 ITEMS), in registers between the
// minimum and the subtraction, so every topology element is read and every entry evaluated once.
// ITEMS == 0: any N; the entries are evaluated a second time after the minimum is known (the second read of
// the topology rows comes from L2).
template <int ITEMS>
__global__ void __launch_bounds__(256) distance_matrix_kernel(const float* __restrict__ top, int pitch,
                                                              const float* __restrict__ ls,
                                                              const int64_t* __restrict__ row,
                                                              const uint8_t* __restrict__ interp_flag,
                                                              const double* __restrict__ wl_all,
                                                              const double* __restrict__ wr_all,
                                                              const float* __restrict__ e_pn_all,
                                                              const float* __restrict__ e_np_all, int N,
                                                              float* __restrict__ matrix)
{
  __shared__ float smin[4];
  const int n = blockIdx.x;
  const long long r0 = row[n];
  const bool interp = interp_flag[n] != 0;
  const float* __restrict__ tp = top + r0 * pitch;
  const float* __restrict__ tn = tp + (interp ? pitch : 0);
  const float ls_prev = ls[r0];
  const float ls_next = interp ? ls[r0 + 1] : ls_prev;
  const double wl = interp ? wl_all[n] : 0.0, wr = interp ? wr_all[n] : 0.0;
  const float e_pn = interp ? e_pn_all[n] : 1.0f, e_np = interp ? e_np_all[n] : 1.0f;
  const bool prev_le_next = ls_prev <= ls_next;
  float mn = __builtin_inff();
  float v[ITEMS > 0 ? ITEMS : 1];
  if constexpr (ITEMS > 0) {
    // loads first (they are independent), then the arithmetic
    float a[ITEMS], b[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const int m = threadIdx.x + 256 * i;
      a[i] = m < N ? tp[m] : 1.0f;
    }
    if (interp) {
#pragma unroll
      for (int i = 0; i < ITEMS; i++) {
        const int m = threadIdx.x + 256 * i;
        b[i] = m < N ? tn[m] : 1.0f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < ITEMS; i++) b[i] = a[i];
    }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const int m = threadIdx.x + 256 * i;
      v[i] = distance_entry(interp, prev_le_next, a[i], b[i], wl, wr, e_pn, e_np, ls_prev, ls_next);
      if (m < N && v[i] < mn) mn = v[i];
    }
  } else {
    for (int m = threadIdx.x; m < N; m += 256) {
      const float x = distance_entry(interp, prev_le_next, tp[m], tn[m], wl, wr, e_pn, e_np, ls_prev, ls_next);
      if (x < mn) mn = x;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float other = __shfl_xor(mn, o, 64);
    if (other < mn) mn = other;
  }
  if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = mn;
  __syncthreads();
  mn = smin[0];
#pragma unroll
  for (int i = 1; i < 4; i++)
    if (smin[i] < mn) mn = smin[i];
  float* __restrict__ out = matrix + (long long)n * N;
  if constexpr (ITEMS > 0) {
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const int m = threadIdx.x + 256 * i;
      if (m < N) out[m] = (m == n) ? 0.0f : __fsub_rn(v[i], mn);
    }
  } else {
    for (int m = threadIdx.x; m < N; m += 256) {
      const float x = distance_entry(interp, prev_le_next, tp[m], tn[m], wl, wr, e_pn, e_np, ls_prev, ls_next);
      out[m] = (m == n) ? 0.0f : __fsub_rn(x, mn);
    }
  }
}

// ------------------------------------------------------------------ packing
// chunk_<c>.hap rows ('0'/'1' bytes, src/data.cpp:541-550) -> bit rows.  One wave per 8 output words (512
// haplotypes of one site): every lane takes 8 bytes with aligned 8-byte loads (two, shifted together, when the
// row does not start on an 8-byte boundary), squeezes their low bits into one byte, and the 64 bytes of the
// wave are the 8 words, stored coalesced.  row_stride is a multiple of 8 words, so segments never straddle rows.
__global__ void __launch_bounds__(256) pack_rows_kernel(const uint8_t* __restrict__ seq, uint64_t* __restrict__ rows,
                                                        int N, int L, int row_stride)
{
  const int lane = lane_id();
  const int segs_per_row = row_stride >> 3;
  const long long nseg = (long long)L * segs_per_row;
  const long long total = (long long)L * N;  // bytes of seq
  const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long seg = wave0; seg < nseg; seg += nwaves) {
    const long long s = seg / segs_per_row;
    const int part = (int)(seg - s * segs_per_row);
    const int n0 = part * 512 + lane * 8;  // first haplotype of this lane
    uint64_t x = 0;
    if (n0 < N) {
      const long long b = s * N + n0;  // first byte wanted
      const uintptr_t addr = reinterpret_cast<uintptr_t>(seq) + (uintptr_t)b;
      const unsigned sh = (unsigned)(addr & 7u) * 8u;
      const uint64_t* q = reinterpret_cast<const uint64_t*>(addr & ~(uintptr_t)7);
      // an aligned word is read only if its first wanted byte lies inside seq (it then shares a page with it)
      x = q[0] >> sh;
      if (sh != 0 && b + (8 - (long long)(sh >> 3)) < total) x |= q[1] << (64u - sh);
    }
    // bit i of `bits` = low bit of byte i ('1' = 0x31); haplotypes past N carry a set bit: a state that never
    // mismatches (the recursion kernels rely on it; the transpose never looks at them)
    unsigned bits = (unsigned)(((x & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
    const int valid = N - n0;  // haplotypes of this lane inside the row
    if (valid < 8) bits |= valid <= 0 ? 0xFFu : (0xFFu << valid) & 0xFFu;
    reinterpret_cast<uint8_t*>(rows)[seg * 64 + lane] = (uint8_t)bits;
  }
}

// 64x64 bit-tile transpose: lane l reads word `wi` of row 64*sblk+l; ballot b
// collects bit b of all 64 rows = 64 consecutive sites of haplotype 64*wi+b.
__global__ void __launch_bounds__(256) transpose_kernel(const uint64_t* __restrict__ rows, uint64_t* __restrict__ cols,
                                                        int N, int L, int row_stride, int col_stride)
{
  const int lane = lane_id();
  const int wi = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (wi * 64 >= N) return;
  const int sblk = blockIdx.x;
  const long long s = (long long)sblk * 64 + lane;
  const uint64_t x = (s < L) ? rows[s * row_stride + wi] : 0ull;
  uint64_t mine = 0;
#pragma unroll
  for (int b = 0; b < 64; b++) {
    const unsigned long long m = __ballot((x >> b) & 1ull);
    if (lane == b) mine = m;
  }
  const int n = wi * 64 + lane;
  if (n < N) cols[(long long)n * col_stride + sblk] = mine;
}

// Column word sw of target k with the forced sites applied: site 0 and site
// L-1 count as derived whatever the haplotype carries, sites past L-1 do not
// exist (src/fast_painting.cpp:65,69,106: `snp != last_snp` ends every skip).
__device__ __forceinline__ uint64_t derived_word(const uint64_t* __restrict__ col, int sw, int L)
{
  const int last = L - 1, lw = last >> 6, lb = last & 63;
  uint64_t x = 0;
  if (sw <= lw) x = col[sw];
  if (sw == lw) x = (x & ((1ull << lb) - 1ull)) | (1ull << lb);
  if (sw == 0) x |= 1ull;
  if (sw > lw) x = 0;
  return x;
}

__global__ void __launch_bounds__(256) count_derived_kernel(const uint64_t* __restrict__ cols, int col_stride, int L,
                                                            int k_begin, int K, int32_t* __restrict__ nd)
{
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= K) return;
  const uint64_t* col = cols + (long long)(k_begin + t) * col_stride;
  int cnt = 0;
  for (int sw = lane_id(); sw < col_stride; sw += 64) cnt += __popcll(derived_word(col, sw, L));
  cnt = wave_sum_int(cnt);
  if (lane_id() == 0) nd[t] = cnt;
}

// One wave per target; each pass covers 64 column words (4096 sites): lane l
// expands the set bits of its word after an exclusive prefix sum of popcounts.
__global__ void __launch_bounds__(256) fill_sites_kernel(const uint64_t* __restrict__ cols, int col_stride, int L,
                                                         int k_begin, int K, const int64_t* __restrict__ off,
                                                         int32_t* __restrict__ dk)
{
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= K) return;
  const int lane = lane_id();
  const uint64_t* col = cols + (long long)(k_begin + t) * col_stride;
  int32_t* dst = dk + off[t];
  int base = 0;
  for (int sw0 = 0; sw0 < col_stride; sw0 += 64) {
    const int sw = sw0 + lane;
    uint64_t x = (sw < col_stride) ? derived_word(col, sw, L) : 0ull;
    const int c = __popcll(x);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    int pos = base + incl - c;
    while (x) {
      const int b = __ffsll((long long)x) - 1;
      dst[pos++] = sw * 64 + b;
      x &= x - 1;
    }
    base += __shfl(incl, 63, 64);
  }
}

// Per-gap terms (src/fast_painting.cpp:66-157): the recombination distance of
// gap j is accumulated left to right starting from r[dk[j]]; the last entry is
// r[L-1] alone.  Stored as coef = r_prob/((1-r_prob)(N-1)), the only form the
// recursion uses (:274,:365,:469,:569), and nor_x_theta.
//
// Two phases per block of 1024 sites.  Summing: a thread walks the SNPs of kSitesPerThread consecutive sites,
// whose gaps are one contiguous run of r, and drops each sum into LDS when the run crosses into the next site.
// The gaps of a panel of thousands are a few SNPs on average and tens at the longest, and a wave moves at the
// pace of its longest lane: over four gaps per lane the longest lane is about twice the mean, over one gap
// four to five times.  Then one site per thread again for exp, division and the record.
constexpr int kSitesPerThread = 4;
constexpr int kSitesPerBlock = 256 * kSitesPerThread;
__global__ void __launch_bounds__(256) site_terms_kernel(const int32_t* __restrict__ dk, const int64_t* __restrict__ off,
                                                         const int32_t* __restrict__ nd, const double* __restrict__ r,
                                                         double log_ntheta, double log_small, double Nm1,
                                                         double* __restrict__ coef, double* __restrict__ nx,
                                                         SiteRec* __restrict__ rec)
{
  const int t = blockIdx.y;
  const int n = nd[t];
  const long long o = off[t];
  constexpr int T = kSitesPerThread, BS = kSitesPerBlock;
  // the 24-byte records of a block are contiguous: they go through LDS and leave as coalesced 8-byte stores
  __shared__ double staged[BS * 3];
  // list entries j-2 .. j+2 of the block's sites, indices clamped to the list: one coalesced load for all of them
  __shared__ int32_t entries[BS + 4];
  __shared__ double gap_sum[BS];
  static_assert(sizeof(SiteRec) == 24, "three 8-byte words per record");
  for (int jb = blockIdx.x * BS; jb < n; jb += gridDim.x * BS) {
    for (int i = threadIdx.x; i < BS + 4; i += blockDim.x) {
      int q = jb - 2 + i;
      q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
      entries[i] = dk[o + q];
    }
    __syncthreads();
    // ---- sums of the gaps of sites j0 .. j0 + cnt - 1, left to right as the reference adds them (:84-96) ----
    const int j0 = jb + (int)threadIdx.x * T;
    if (j0 < n) {
      const int cnt = n - j0 < T ? n - j0 : T;
      const int e0 = (int)threadIdx.x * T + 2;  // entries[e0 + i] = first SNP of site j0 + i
      // one past the last SNP of site j0 + i: the next site's first, or for the last site of the list its own + 1
      auto gap_end = [&](int i) { return j0 + i + 1 < n ? entries[e0 + i + 1] : entries[e0 + i] + 1; };
      const int s_first = entries[e0];
      const int s_stop = gap_end(cnt - 1);
      int site = 0, end = gap_end(0);
      double rsum = r[s_first];
      // the loads of 8 terms go out together: one round trip to memory instead of eight
      for (int s = s_first + 1; s < s_stop; s += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (s + u < s_stop) ? r[s + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (s + u < s_stop) {
            if (s + u == end) {  // the run crosses into the next site: its sum starts with this term
              gap_sum[threadIdx.x * T + site] = rsum;
              site++;
              end = gap_end(site);
              rsum = v[u];
            } else {
              rsum += v[u];
            }
          }
        }
      }
      gap_sum[threadIdx.x * T + site] = rsum;  // (site == cnt - 1: the loop has passed the first SNP of every later site)
    }
    __syncthreads();
    // ---- one site per thread: the terms and the record ----
    for (int k = 0; k < T; k++) {
      const int jl = k * (int)blockDim.x + (int)threadIdx.x;
      const int j = jb + jl;
      if (j < n) {
        const double rsum = gap_sum[jl];
        double nxv = -rsum + log_ntheta;
        double rp = 1.0 - exp(-rsum);
        if (rp > 0.99) {
          rp = 0.99;
          nxv = log_small + log_ntheta;
        }
        const double cf = rp / ((1.0 - rp) * Nm1);
        if (coef) {
          coef[o + j] = cf;
          nx[o + j] = nxv;
        }
        if (rec) {
          const int f2 = entries[jl + 4];  // dk[min(j + 2, n - 1)]
          const int b2 = entries[jl];      // dk[max(j - 2, 0)]
          staged[3 * jl] = cf;
          staged[3 * jl + 1] = nxv;
          staged[3 * jl + 2] = __hiloint2double(b2, f2);  // dk_fwd2 in the low word, dk_bwd2 in the high word
        }
      }
    }
    __syncthreads();  // (also: every thread is done with `entries` and `gap_sum` before the next block of sites)
    if (rec) {
      const int cnt3 = 3 * (n - jb < BS ? n - jb : BS);
      double* dst = reinterpret_cast<double*>(rec + o + jb);
      for (int i = threadIdx.x; i < cnt3; i += blockDim.x) dst[i] = staged[i];
      __syncthreads();
    }
  }
}

__global__ void section_last_terms_kernel(const int32_t* __restrict__ dk, const int64_t* __restrict__ off,
                                          const int32_t* __restrict__ bej, int K, int W, int w,
                                          const double* __restrict__ r, double log_ntheta, double log_small, double Nm1,
                                          double* __restrict__ coef, double* __restrict__ nx)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= K) return;
  const int s = dk[off[t] + bej[(long long)t * W + w]];
  const double rsum = r[s];
  double nxv = -rsum + log_ntheta;
  double rp = 1.0 - exp(-rsum);
  if (rp > 0.99) {
    rp = 0.99;
    nxv = log_small + log_ntheta;
  }
  coef[t] = rp / ((1.0 - rp) * Nm1);
  nx[t] = nxv;
}

// Window bookkeeping (src/fast_painting.cpp:33-45,:73-82,:111-120,:163-166) in
// closed form: for the edge e = wb[w] (w >= 1), q = first derived site >= e
// closes window w-1 and its predecessor opens window w.
__global__ void window_stones_kernel(const int32_t* __restrict__ dk, const int64_t* __restrict__ off,
                                     const int32_t* __restrict__ nd, int K, int W, const int32_t* __restrict__ wb,
                                     int32_t* __restrict__ bbj, int32_t* __restrict__ bej)
{
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)K * W) return;
  const int t = (int)(idx / W), w = (int)(idx - (long long)t * W);
  const int32_t* d = dk + off[t];
  const int n = nd[t];
  if (w == 0) bbj[(long long)t * W] = 0;
  if (w == W - 1) bej[(long long)t * W + w] = n - 1;
  if (w >= 1) {
    const int e = wb[w];
    int lo = 0, hi = n - 1;  // d[n-1] = L-1 >= e always
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (d[mid] >= e) hi = mid;
      else lo = mid + 1;
    }
    bej[(long long)t * W + w - 1] = lo;
    bbj[(long long)t * W + w] = lo - 1;
  }
}

// Plain-HIP recursion (variant 0).
template <int MPL, int G>
__global__ void __launch_bounds__(64 * G) paint_kernel(
    const uint64_t* __restrict__ rows, const int32_t* __restrict__ dk_all, const double* __restrict__ coef_all,
    const double* __restrict__ nx_all, const int64_t* __restrict__ off, const int32_t* __restrict__ nd_all,
    const int32_t* __restrict__ bbj_all, const int32_t* __restrict__ bej_all, const int32_t* __restrict__ order,
    const int32_t* __restrict__ wb, uint8_t* __restrict__ out, const Scal p)
{
  __shared__ double red[2][G];
  int slot = 0;
  const int t = order[blockIdx.x];
  const int k = p.k_begin + t;
  const int lane = lane_id();
  const int wbase = (threadIdx.x >> 6) * MPL;
  const int N = p.N, W = p.W, stride = p.row_stride;
  const int nd = nd_all[t];
  const int32_t* __restrict__ dk = dk_all + off[t];
  const double* __restrict__ coef = coef_all + off[t];
  const double* __restrict__ nx = nx_all + off[t];
  const int32_t* __restrict__ bbj = bbj_all + (long long)t * W;
  const int32_t* __restrict__ bej = bej_all + (long long)t * W;
  const size_t R = 56 + (size_t)8 * N;
  const int kw = k >> 6, kb = k & 63;
  const bool first_thread = threadIdx.x == 0;
  const PaintConsts& c = p.c;

  double a[MPL];

  // ---------------- forward (src/fast_painting.cpp:214-408) ----------------
  {
    const uint64_t* row = rows;  // site 0
    const bool sk = (row_word(row, kw, stride) >> kb) & 1ull;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < MPL; i++) {
      const int wi = wbase + i;
      const uint64_t w = row_word(row, wi, stride);
      const uint64_t vm = valid_mask(wi, N, k);
      const uint64_t mm = sk ? (~w & vm) : 0ull;
      const bool v = (vm >> lane) & 1ull, m = (mm >> lane) & 1ull;
      a[i] = v ? (m ? c.prior_mis : c.prior_match) : 0.0;
      acc += a[i];
    }
    double S = block_sum<G>(acc, red, slot);
    double ls = 0.0;
    int wcur = 0;
    auto emit = [&](int j, int s) {
      while (wcur < W && bbj[wcur] == j) {
        uint8_t* rec = out + ((size_t)wcur * p.K + t) * R;
        if (first_thread) {
          put_u32(rec, (uint32_t)wb[wcur]);
          put_u32(rec + 4, (uint32_t)(wb[wcur + 1] - 1));
          put_block_header(rec + 8, N, s, (float)ls);
        }
        store_payload<MPL>(reinterpret_cast<float*>(rec + 32), a, wbase, N);
        wcur++;
      }
    };
    emit(0, 0);
    double rx = coef[0] * S;
    for (int j = 1; j < nd; j++) {
      const int s = dk[j];
      row = rows + (size_t)s * stride;
      const bool skj = (row_word(row, kw, stride) >> kb) & 1ull;
      ls += nx[j - 1];
      acc = 0.0;
#pragma unroll
      for (int i = 0; i < MPL; i++) {
        const int wi = wbase + i;
        const uint64_t w = row_word(row, wi, stride);
        const uint64_t vm = valid_mask(wi, N, k);
        const uint64_t mm = skj ? (~w & vm) : 0ull;
        const bool v = (vm >> lane) & 1ull, m = (mm >> lane) & 1ull;
        double x = a[i] + rx;
        const double xm = x * c.rho_theta;
        x = m ? xm : x;
        a[i] = v ? x : 0.0;
        acc += a[i];
      }
      S = block_sum<G>(acc, red, slot);
      if (S < c.lo || S > c.hi) {
#pragma unroll
        for (int i = 0; i < MPL; i++) a[i] = a[i] / S;
        ls += log(S);
        S = 1.0;
      }
      rx = S * coef[j];
      emit(j, s);
    }
  }

  // ---------------- backward (src/fast_painting.cpp:410-599) ----------------
  {
    double ls = c.log_Nm1 - (double)nd * c.log_ntheta;
    const uint64_t* row_next = rows + (size_t)(p.L - 1) * stride;  // row of the site handled one step earlier
    bool sk_next = (row_word(row_next, kw, stride) >> kb) & 1ull;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < MPL; i++) {
      const int wi = wbase + i;
      const uint64_t w = row_word(row_next, wi, stride);
      const uint64_t vm = valid_mask(wi, N, k);
      const uint64_t mm = sk_next ? (~w & vm) : 0ull;
      const bool v = (vm >> lane) & 1ull, m = (mm >> lane) & 1ull;
      a[i] = v ? 1.0 : 0.0;
      acc += v ? (m ? c.theta_d : c.ntheta_d) : 0.0;
    }
    double B = block_sum<G>(acc, red, slot);
    int wcur = W - 1;
    // stones taken at the last SNP hold beta = 1 for every state, the target included (:427-432,:447-462)
    while (wcur >= 0 && bej[wcur] == nd - 1) {
      uint8_t* rec = out + ((size_t)wcur * p.K + t) * R;
      if (first_thread) put_block_header(rec + 32 + (size_t)4 * N, N, p.L - 1, (float)ls);
      float* dst = reinterpret_cast<float*>(rec + 56 + (size_t)4 * N);
#pragma unroll
      for (int i = 0; i < MPL; i++) {
        const int n = (wbase + i) * 64 + lane;
        if (n < N) dst[n] = 1.0f;
      }
      wcur--;
    }
    double rx = coef[nd - 1] * B;
    for (int j = nd - 2; j >= 0; j--) {
      const int s = dk[j];
      const uint64_t* row = rows + (size_t)s * stride;
      const bool sk = (row_word(row, kw, stride) >> kb) & 1ull;
      ls += nx[j + 1];
      const double add_all = rx / c.ntheta_d;
      const double add_mis = rx / c.theta_d - add_all;
      acc = 0.0;
#pragma unroll
      for (int i = 0; i < MPL; i++) {
        const int wi = wbase + i;
        const uint64_t vm = valid_mask(wi, N, k);
        const uint64_t wn = row_word(row_next, wi, stride);
        const uint64_t wc = row_word(row, wi, stride);
        const uint64_t mn = sk_next ? (~wn & vm) : 0ull;
        const uint64_t mc = sk ? (~wc & vm) : 0ull;
        const bool v = (vm >> lane) & 1ull, m1 = (mn >> lane) & 1ull, m0 = (mc >> lane) & 1ull;
        double x = a[i];
        const double x1 = x + add_mis;
        x = m1 ? x1 : x;
        x = x + add_all;
        const double x2 = x * c.rho_theta;
        x = m1 ? x2 : x;
        a[i] = v ? x : 0.0;
        acc += (m0 ? c.theta_d : c.ntheta_d) * a[i];
      }
      B = block_sum<G>(acc, red, slot);
      if (B < c.lo || B > c.hi) {
#pragma unroll
        for (int i = 0; i < MPL; i++) a[i] = a[i] / B;
        ls += (double)fast_log_dev((float)B);  // float approximation on this side (:562)
        B = 1.0;
      }
      rx = B * coef[j];
      while (wcur >= 0 && bej[wcur] == j) {
        uint8_t* rec = out + ((size_t)wcur * p.K + t) * R;
        if (first_thread) put_block_header(rec + 32 + (size_t)4 * N, N, s, (float)ls);
        store_payload<MPL>(reinterpret_cast<float*>(rec + 56 + (size_t)4 * N), a, wbase, N);
        wcur--;
      }
      row_next = row;
      sk_next = sk;
    }
  }
}


template <int MPL, int G>
hipError_t launch_plain_t(const PaintArgs& a, hipStream_t st)
{
  Scal s;
  s.row_stride = a.row_stride;
  s.N = a.N;
  s.L = a.L;
  s.W = a.W;
  s.k_begin = a.k_begin;
  s.K = a.K;
  s.c = a.c;
  hipLaunchKernelGGL((paint_kernel<MPL, G>), dim3(a.K), dim3(64 * G), 0, st, a.rows, a.dk, a.coef, a.nx, a.off, a.nd,
                     a.bbj, a.bej, a.order, a.wb, a.out, s);
  return hipGetLastError();
}

}  // namespace

// ------------------------------------------------------------------ host side
// Plain-HIP kernels exist for a few small geometries only (cross-check of the register-resident ones).
static const int kMplPlain[] = {1, 4, 16};
static const int kMplRegs[] = {4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80};

// Words per lane of the register-resident kernel with G waves for M words per row (0: none)
static int regs_words_per_lane(int M, int G)
{
  const int need = (M + G - 1) / G;
  for (int m : kMplRegs)
    if (m >= need) return (G == 8 && m == 80) ? 0 : m;  // 8 x 80: stone staging would exceed the LDS of a CU
  return 0;
}

// Words per packed row: a multiple of 8 that holds the N bits and, beyond 80 words, all G x MPL words that the
// narrowest kernel geometry reads of a row (every word it reads must carry the padding that pack_rows writes,
// set bits, i.e. states that never mismatch: 88 words of bits take 96, as 4 x 24, 2 x 48 or 8 x 12).
int row_stride_words(int N)
{
  const int M = (N + 63) / 64;
  int stride = (M + 7) / 8 * 8;
  if (M > 80) {
    int narrowest = 0;
    for (int G = 2; G <= 8; G *= 2) {
      const int mpl = regs_words_per_lane(M, G);
      if (mpl > 0 && (narrowest == 0 || G * mpl < narrowest)) narrowest = G * mpl;
    }
    if (narrowest > stride) stride = (narrowest + 7) / 8 * 8;
  }
  return stride;
}
int col_stride_words(int L) { return (L + 63) / 64; }

PaintGeometry choose_geometry(int N, int prefer_waves, int variant)
{
  const int M = (N + 63) / 64;
  const int maxG = variant == 0 ? 4 : 8;
  int G0 = 1;
  if (prefer_waves == 2 || prefer_waves == 4 || prefer_waves == 8) G0 = prefer_waves;
  // The register-resident kernel is fastest with at most 32 states per lane (120 VGPRs: four waves per
  // SIMD), but the eighth wave of a target costs more than it brings: measured on MI355X, N=2000 is best with
  // 1 wave per target (6.6 ms), N=5008 with 4 x 20 words (97 ms; 2 x 40: 103), N=10000 with 4 x 40 (231 ms;
  // 8 x 20: 284).
  if (variant >= 1 && prefer_waves == 0)
    while (G0 < maxG && (M + G0 - 1) / G0 > (G0 >= 4 ? 40 : 32)) G0 *= 2;
  if (G0 > maxG) G0 = maxG;
  // words per lane for G waves, 0 if that wave count cannot paint N states
  auto fit = [&](int G) {
    if (variant == 0) {
      const int need = (M + G - 1) / G;
      for (int m : kMplPlain)
        if (m >= need) return m;
      return 0;
    }
    const int mpl = regs_words_per_lane(M, G);
    // the register-resident kernel reads G*MPL words of every row and needs the padding bits of all of them
    return (variant >= 1 && G * mpl > row_stride_words(N)) ? 0 : mpl;
  };
  // the wave count asked for or given by the rule; failing that fewer waves, then more
  for (int G = G0; G >= 1; G /= 2)
    if (const int mpl = fit(G)) return {G, mpl};
  for (int G = 2 * G0; G <= maxG; G *= 2)
    if (const int mpl = fit(G)) return {G, mpl};
  return {G0, 0};  // MPL == 0: N too large for this variant
}

// Geometry for a launch of K targets: the wave count that a model of the launch prices lowest (the rule of
// choose_geometry where the model sees no difference, and beyond the panel sizes it was fitted on).  All targets
// of a chunk take about the same time, so a launch runs in rounds: 2 K G waves (a forward and a backward workgroup
// per target) over 1024 SIMDs x the waves per SIMD that registers and LDS allow (waves_per_cu), fewer if the launch
// does not fill the chip.
// Time per site of a round = the larger of
//   issue:   waves per SIMD x instructions per site x ns per issue slot, with 3 instructions per word + 24
//            (+ 10, 14, 21 for the exchange between 2, 4, 8 waves) and 3.45, 2.73, 2.24, 2.09, 1.89 ns per slot
//            at 2, 3, 4, 5, 6 or more waves per SIMD;
//   latency: the dependent chain of one site, 187 + 14.8 ns per word for one wave, 227 + 20.6, 359 + 6.8, 460 + 8.8
//            for 2, 4, 8 waves (what a launch of a few targets takes per site).
// A last round that is not full runs at the occupancy its waves reach, but never under 0.86 of a full round: its
// waves start beside the last ones of the round before and share their SIMDs with them.
// The exchange, latency and last-round constants are fits to 240 launches of the present kernels on MI355X
// (N = 1000 ... 5008, K = N / 16 ... N, every wave count; tools/geometry_sweep_k.py), the issue slots to those of the
// first kernels.  What the size rule does not see is the number of rounds and the fill: N = K = 3000 is 2.93
// rounds of 2 x 24 words (17.2 ms) but 1.95 rounds of 1 x 48 (15.5 ms); 375 of those targets take 4.2 ms as 2 x 24
// and 3.2 ms as 4 x 12.  Over the measured launches the rule is 7.9 % (geometric mean) behind the best wave count,
// the model's choice 0.5 % (worst: 13.6 % at N = 2000, K = 500), and it is the best count for every K = N.
namespace {

// Waves of the kernel that a CU holds: 512 VGPRs per SIMD for 56 + 2 MPL registers per lane, and 160 KiB of LDS,
// handed out in units of 1280 bytes, for the stone staging of every workgroup.  Measured: one round takes 2048
// targets of 1 x 36 words (16 workgroups per CU, 10 240 bytes each: the LDS exactly), 1536 of 1 x 48, but 1280
// of 1 x 56 (3 waves per SIMD by registers, 10 workgroups per CU by LDS).
int waves_per_cu(PaintGeometry g)
{
  const int by_regs = 4 * std::max(1, std::min(8, 512 / ((56 + 2 * g.MPL + 7) / 8 * 8)));
  const int lds_wg = (g.G * g.MPL * 256 + 24 * g.G + 1279) / 1280 * 1280;
  const int by_lds = g.G * (163840 / lds_wg);
  return std::max(g.G, std::min(by_regs, by_lds));
}

// ns per site of one round of waves at `occ` waves per SIMD
double round_cost(PaintGeometry g, int occ)
{
  static const double ns_per_slot[9] = {0.0, 4.0, 3.45, 2.73, 2.24, 2.09, 1.89, 1.89, 1.89};
  const double exchange = g.G == 1 ? 0.0 : g.G == 2 ? 10.0 : g.G == 4 ? 14.0 : 21.0;
  const double issue = occ * (3.0 * g.MPL + 24.0 + exchange) * ns_per_slot[occ];
  const double latency = g.G == 1   ? 187.0 + 14.8 * g.MPL
                         : g.G == 2 ? 227.0 + 20.6 * g.MPL
                         : g.G == 4 ? 359.0 + 6.8 * g.MPL
                                    : 460.0 + 8.8 * g.MPL;
  return std::max(issue, latency);
}

// ns per site of a launch of K targets
double launch_cost(PaintGeometry g, int K)
{
  const long long waves = 2LL * K * g.G;
  const int cu = waves_per_cu(g);
  // workgroups do not spread evenly over a chip they do not fill: 768 waves already put two on some SIMDs
  const int occ = (int)std::max(1LL, std::min<long long>((cu + 3) / 4, (waves + 767) / 768));
  const long long slots = std::min(256LL * cu, 1024LL * occ);
  const long long rounds = (waves + slots - 1) / slots;
  // the last round runs at the occupancy its waves reach, beside the end of the round before (N = K = 2100 as
  // 1 x 36: 13.4 ms, of which the 52 targets of the second round take 5)
  const long long rest = waves - (rounds - 1) * slots;
  const int occ_last = (int)std::max(1LL, std::min<long long>(occ, (rest + 767) / 768));
  double last = round_cost(g, occ_last);
  if (rounds > 1) last = std::max(last, 0.86 * round_cost(g, occ));
  return (double)(rounds - 1) * round_cost(g, occ) + last;
}

}  // namespace

PaintGeometry choose_geometry_for_launch(int N, int K, int prefer_waves, int variant)
{
  const PaintGeometry rule = choose_geometry(N, prefer_waves, variant);
  if (variant < 1 || prefer_waves != 0 || rule.MPL <= 0 || K <= 0) return rule;
  if (N > 5120) return rule;  // beyond the sizes the constants were fitted on
  PaintGeometry best = rule;
  double best_cost = launch_cost(rule, K);
  for (int G = 1; G <= 8; G *= 2) {
    const PaintGeometry g = choose_geometry(N, G, variant);  // a nonzero wave count is taken as given
    if (g.MPL <= 0 || g.G != G) continue;
    const double c = launch_cost(g, K);
    if (c < best_cost) {
      best = g;
      best_cost = c;
    }
  }
  return best;
}

// Bytes to allocate for the packed rows: the recursion kernel fetches whole 4-word chunks for all
// G*MPL words of a wave group, which may run past the end of the last row.
size_t rows_alloc_bytes(int N, int L) { return ((size_t)L * row_stride_words(N) + 8 * 80 + 64) * sizeof(uint64_t); }

hipError_t launch_pack_rows(const uint8_t* d_seq, uint64_t* d_rows, int N, int L, int row_stride, hipStream_t st)
{
  const long long segs = (long long)L * (row_stride / 8);
  long long blocks = (segs + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_seq, d_rows, N, L, row_stride);
  return hipGetLastError();
}

hipError_t launch_transpose(const uint64_t* d_rows, uint64_t* d_cols, int N, int L, int row_stride, int col_stride,
                            hipStream_t st)
{
  const int M = (N + 63) / 64;
  hipLaunchKernelGGL(transpose_kernel, dim3(col_stride, (M + 3) / 4), dim3(256), 0, st, d_rows, d_cols, N, L,
                     row_stride, col_stride);
  return hipGetLastError();
}

hipError_t launch_count_derived(const uint64_t* d_cols, int col_stride, int L, int k_begin, int K, int32_t* d_nd,
                                hipStream_t st)
{
  hipLaunchKernelGGL(count_derived_kernel, dim3((K + 3) / 4), dim3(256), 0, st, d_cols, col_stride, L, k_begin, K, d_nd);
  return hipGetLastError();
}

hipError_t launch_fill_sites(const uint64_t* d_cols, int col_stride, int L, int k_begin, int K, const int64_t* d_off,
                             int32_t* d_dk, hipStream_t st)
{
  hipLaunchKernelGGL(fill_sites_kernel, dim3((K + 3) / 4), dim3(256), 0, st, d_cols, col_stride, L, k_begin, K, d_off,
                     d_dk);
  return hipGetLastError();
}

hipError_t launch_site_terms(const int32_t* d_dk, const int64_t* d_off, const int32_t* d_nd, int K, int max_nd,
                             const double* d_r, int L, const PaintConsts& c, double* d_coef, double* d_nx,
                             SiteRec* d_rec, hipStream_t st)
{
  (void)L;
  int bx = (max_nd + kSitesPerBlock - 1) / kSitesPerBlock;
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(site_terms_kernel, dim3(bx, K), dim3(256), 0, st, d_dk, d_off, d_nd, d_r, c.log_ntheta,
                     c.log_small, c.Nm1, d_coef, d_nx, d_rec);
  return hipGetLastError();
}

hipError_t launch_section_last_terms(const int32_t* d_dk, const int64_t* d_off, const int32_t* d_bej, int K, int W, int w,
                                     const double* d_r, const PaintConsts& c, double* d_coef, double* d_nx,
                                     hipStream_t st)
{
  hipLaunchKernelGGL(section_last_terms_kernel, dim3((K + 255) / 256), dim3(256), 0, st, d_dk, d_off, d_bej, K, W, w, d_r,
                     c.log_ntheta, c.log_small, c.Nm1, d_coef, d_nx);
  return hipGetLastError();
}

hipError_t launch_window_stones(const int32_t* d_dk, const int64_t* d_off, const int32_t* d_nd, int K, int W,
                                const int32_t* d_wb, int32_t* d_bbj, int32_t* d_bej, hipStream_t st)
{
  const long long n = (long long)K * W;
  hipLaunchKernelGGL(window_stones_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_dk, d_off, d_nd, K,
                     W, d_wb, d_bbj, d_bej);
  return hipGetLastError();
}

hipError_t launch_gather_logscales(const float* d_ls, const int64_t* d_row, const uint8_t* d_interp, int N, float* d_out,
                                   hipStream_t st)
{
  hipLaunchKernelGGL(gather_logscales_kernel, dim3((N + 255) / 256), dim3(256), 0, st, d_ls, d_row, d_interp, N, d_out);
  return hipGetLastError();
}

hipError_t launch_distance_matrix(const float* d_top, int pitch, const float* d_ls, const int64_t* d_row,
                                  const uint8_t* d_interp, const double* d_wl, const double* d_wr, const float* d_epn,
                                  const float* d_enp, int N, float* d_matrix, hipStream_t st)
{
#define RELATE_DM(I)                                                                                                   \
  hipLaunchKernelGGL(distance_matrix_kernel<I>, dim3(N), dim3(256), 0, st, d_top, pitch, d_ls, d_row, d_interp, d_wl, \
                     d_wr, d_epn, d_enp, N, d_matrix)
  if (N <= 256 * 4) RELATE_DM(4);
  else if (N <= 256 * 8) RELATE_DM(8);
  else if (N <= 256 * 12) RELATE_DM(12);
  else if (N <= 256 * 20) RELATE_DM(20);
  else if (N <= 256 * 40) RELATE_DM(40);
  else RELATE_DM(0);
#undef RELATE_DM
  return hipGetLastError();
}

// MinMatch::Initialize's scans (src/tree_builder.cpp:57-127): row minima, then candidate pairs counted per row
// and their offsets.  d_offsets[N] is the number of pairs.
hipError_t launch_minmatch_scan(const float* d_matrix, int N, float threshold, float* d_min_values, int32_t* d_counts,
                                int64_t* d_offsets, hipStream_t st)
{
  if (N < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(minmatch::row_min_kernel, dim3(N), dim3(64), 0, st, d_matrix, N, threshold, d_min_values);
  hipLaunchKernelGGL(minmatch::count_candidates_kernel, dim3(N), dim3(64), 0, st, d_matrix, d_min_values, N, d_counts);
  hipLaunchKernelGGL(minmatch::scan_counts_kernel, dim3(1), dim3(minmatch::kScanThreads), 0, st, d_counts, N, d_offsets);
  return hipGetLastError();
}

// The pairs themselves, row by row in ascending j, at the offsets of launch_minmatch_scan; at most `capacity`
// are written.
hipError_t launch_minmatch_fill(const float* d_matrix, int N, const float* d_min_values, const int64_t* d_offsets,
                                int64_t capacity, int32_t* d_ci, int32_t* d_cj, float* d_cd, hipStream_t st)
{
  if (N < 1 || capacity < 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(minmatch::fill_candidates_kernel, dim3(N), dim3(64), 0, st, d_matrix, d_min_values, N, d_offsets,
                     capacity, d_ci, d_cj, d_cd);
  return hipGetLastError();
}

hipError_t launch_fast_log(const float* d_in, float* d_out, int n, hipStream_t st)
{
  hipLaunchKernelGGL(fast_log_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_in, d_out, n);
  return hipGetLastError();
}

// defined in paint_regs_g*.hip
hipError_t launch_paint_regs_g1(const PaintArgs& a, int MPL, hipStream_t st);
hipError_t launch_paint_regs_g2(const PaintArgs& a, int MPL, hipStream_t st);
hipError_t launch_paint_regs_g4(const PaintArgs& a, int MPL, hipStream_t st);
hipError_t launch_paint_regs_g8(const PaintArgs& a, int MPL, hipStream_t st);

#define RELATE_PLAIN_CASE(M, GG) \
  if (g.MPL == M && g.G == GG) return launch_plain_t<M, GG>(a, st);

hipError_t launch_paint(const PaintArgs& a, PaintGeometry g, int variant, hipStream_t st)
{
  if (g.MPL <= 0 || (long long)g.G * g.MPL * 64 < a.N) return hipErrorInvalidValue;
  if (a.row_stride % 8 != 0 || (long long)a.row_stride * 64 < a.N || a.K < 1 || a.W < 1 || a.L < 2 || a.N < 2)
    return hipErrorInvalidValue;
  if (variant == 0) {
    RELATE_PLAIN_CASE(1, 1) RELATE_PLAIN_CASE(4, 1) RELATE_PLAIN_CASE(16, 1)
    RELATE_PLAIN_CASE(1, 2) RELATE_PLAIN_CASE(4, 2) RELATE_PLAIN_CASE(16, 2)
    RELATE_PLAIN_CASE(1, 4) RELATE_PLAIN_CASE(4, 4) RELATE_PLAIN_CASE(16, 4)
    return hipErrorInvalidValue;
  }
  if (variant != 1) return hipErrorInvalidValue;
  if (g.G * g.MPL > a.row_stride || !a.rec) return hipErrorInvalidValue;
  switch (g.G) {
    case 1: return launch_paint_regs_g1(a, g.MPL, st);
    case 2: return launch_paint_regs_g2(a, g.MPL, st);
    case 4: return launch_paint_regs_g4(a, g.MPL, st);
    case 8: return launch_paint_regs_g8(a, g.MPL, st);
    default: return hipErrorInvalidValue;
  }
}

// defined in paint_regs_g*.hip
hipError_t launch_repaint_regs_g1(const RepaintLaunch& a, int MPL, hipStream_t st);
hipError_t launch_repaint_regs_g2(const RepaintLaunch& a, int MPL, hipStream_t st);
hipError_t launch_repaint_regs_g4(const RepaintLaunch& a, int MPL, hipStream_t st);
hipError_t launch_repaint_regs_g8(const RepaintLaunch& a, int MPL, hipStream_t st);

hipError_t launch_repaint(const RepaintLaunch& a, PaintGeometry g, hipStream_t st)
{
  if (g.MPL <= 0 || (long long)g.G * g.MPL * 64 < a.N || g.G * g.MPL > a.row_stride) return hipErrorInvalidValue;
  if (a.pitch != g.G * g.MPL * 64 || a.K < 1 || a.w < 0 || a.w >= a.W || !a.rec) return hipErrorInvalidValue;
  switch (g.G) {
    case 1: return launch_repaint_regs_g1(a, g.MPL, st);
    case 2: return launch_repaint_regs_g2(a, g.MPL, st);
    case 4: return launch_repaint_regs_g4(a, g.MPL, st);
    case 8: return launch_repaint_regs_g8(a, g.MPL, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace relate
