// Device-side entry points of the painting path (launchers around the gfx950
// kernels in paint_kernels.hip).  Internal to the library: the public boundary
// is include/relate_gpu.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace relate {

// Rescale thresholds of the recursions: kRescaleLo and its reciprocal (src/fast_painting.hpp:20-21).
constexpr double kRescaleLo = 1e-10;

// Constants of one painter, computed on the host with the host's libm exactly
// as the reference's constructors do (src/fast_painting.hpp:19-45).
struct PaintConsts {
  double Nm1;           // N - 1
  double prior_mis;     // 1.0*prior_theta + prior_ntheta : alpha_0 of a mismatching state
  double prior_match;   // prior_ntheta                   : alpha_0 of a matching state
  double rho_theta;     // theta_ratio + 1.0              : emission ratio of a mismatching state
  double log_ntheta;    // log(1 - theta_painter)
  double log_small;     // log(0.01)
  double log_Nm1;       // log(N - 1)
  double theta_d;       // Data::theta   (backward pass only)
  double ntheta_d;      // Data::ntheta
  double inv_theta_d;   // 1 / theta_d, 1 / ntheta_d (the register-resident kernel multiplies where the
  double inv_ntheta_d;  // reference divides, src/fast_painting.cpp:488-489)
  double lo, hi;        // rescale thresholds 1e-10, 1e10
  int32_t same_theta;   // theta_painter == theta_data: the backward pass may run on one row per step
  int32_t reserved;
};

// Per derived site j of a target, for the register-resident kernel: the two per-gap terms and the list
// entries two steps ahead in either direction (what a sweep prefetches for the site after the next).
struct SiteRec {
  double coef;      // r_prob/((1-r_prob)(N-1)) of gap j
  double nx;        // nor_x_theta of gap j
  int32_t dk_fwd2;  // dk[min(j+2, nd-1)]
  int32_t dk_bwd2;  // dk[max(j-2, 0)]
};

struct PaintArgs {
  const uint64_t* rows;    // [L][row_stride] bit rows: bit (n&63) of word n>>6 = seq[s][n]=='1'
  int row_stride;          // in 64-bit words, multiple of 8
  const int32_t* dk;       // flat derived-site lists
  const double* coef;      // flat r_prob/((1-r_prob)(N-1)) per derived site
  const double* nx;        // flat nor_x_theta per derived site
  const SiteRec* rec;      // the same interleaved (the register-resident kernel reads this instead of coef / nx)
  const int64_t* off;      // [K] start of target t's lists
  const int32_t* nd;       // [K] number of derived sites
  const int32_t* bbj;      // [K][W] index into the list where window w's alpha stone is taken
  const int32_t* bej;      // [K][W] same for the beta stone
  const int32_t* order;    // [K] launch order (longest target first)
  const int32_t* wb;       // [W+1] window boundaries (device copy)
  uint8_t* out;            // [W][K][56+8N] records
  uint64_t* trace;         // diagnostics (or null): 4 words per workgroup of the register-resident kernel
  int N, L, W, k_begin, K;
  PaintConsts c;
};

// Geometry of the recursion kernel for N states: G waves per target, each lane
// of a wave owns MPL states (words wave*MPL .. wave*MPL+MPL-1 of a bit row).
struct PaintGeometry { int G, MPL; };
// variant as in launch_paint; MPL == 0 in the result means N is out of range for that variant
PaintGeometry choose_geometry(int N, int prefer_waves /*0 = default*/, int variant);
// The same for a launch of K targets (K decides how many rounds of waves the chip runs).
PaintGeometry choose_geometry_for_launch(int N, int K, int prefer_waves /*0 = default*/, int variant);

int row_stride_words(int N);  // words per packed row
size_t rows_alloc_bytes(int N, int L);  // allocation size of the packed rows, slack for chunk over-reads included
int col_stride_words(int L);  // words per packed column

hipError_t launch_pack_rows(const uint8_t* d_seq, uint64_t* d_rows, int N, int L, int row_stride, hipStream_t st);
hipError_t launch_transpose(const uint64_t* d_rows, uint64_t* d_cols, int N, int L, int row_stride, int col_stride,
                            hipStream_t st);
hipError_t launch_count_derived(const uint64_t* d_cols, int col_stride, int L, int k_begin, int K, int32_t* d_nd,
                                hipStream_t st);
hipError_t launch_fill_sites(const uint64_t* d_cols, int col_stride, int L, int k_begin, int K, const int64_t* d_off,
                             int32_t* d_dk, hipStream_t st);
// writes coef / nx (if not null) and rec (if not null)
hipError_t launch_site_terms(const int32_t* d_dk, const int64_t* d_off, const int32_t* d_nd, int K, int max_nd,
                             const double* d_r, int L, const PaintConsts& c, double* d_coef, double* d_nx,
                             SiteRec* d_rec, hipStream_t st);
// Gap terms of the last site of window w's section of every target: r[last] alone (src/fast_painting.cpp:737-750).
hipError_t launch_section_last_terms(const int32_t* d_dk, const int64_t* d_off, const int32_t* d_bej, int K, int W, int w,
                                     const double* d_r, const PaintConsts& c, double* d_coef, double* d_nx,
                                     hipStream_t st);
hipError_t launch_window_stones(const int32_t* d_dk, const int64_t* d_off, const int32_t* d_nd, int K, int W,
                                const int32_t* d_wb, int32_t* d_bbj, int32_t* d_bej, hipStream_t st);
// variant: 0 = plain HIP recursion (cross-check, N <= 4096), 1 = register-resident recursion
hipError_t launch_paint(const PaintArgs& a, PaintGeometry g, int variant, hipStream_t st);
hipError_t launch_fast_log(const float* d_in, float* d_out, int n, hipStream_t st);
// DistanceMeasure::GetMatrix (src/anc_builder.cpp:113-211) on a device-resident topology: row n of `matrix`
// from topology row `row[n]` (and the next one where interp[n] != 0, with weights wl[n], wr[n] and the two
// exponentials of the logscale difference, epn[n] = exp(ls[row] - ls[row+1]), enp[n] = exp(ls[row+1] - ls[row]),
// taken by the host as the reference takes them, :171-172).
hipError_t launch_distance_matrix(const float* d_top, int pitch, const float* d_ls, const int64_t* d_row,
                                  const uint8_t* d_interp, const double* d_wl, const double* d_wr, const float* d_epn,
                                  const float* d_enp, int N, float* d_matrix, hipStream_t st);
// out[2n] = ls[row[n]], out[2n+1] = ls[row[n] + (interp[n] ? 1 : 0)]: what the host needs for those exponentials
hipError_t launch_gather_logscales(const float* d_ls, const int64_t* d_row, const uint8_t* d_interp, int N, float* d_out,
                                   hipStream_t st);

// The scans of MinMatch::Initialize (src/tree_builder.cpp:57-127) over a device-resident N x N matrix:
// d_min_values[N] (threshold added), d_counts[N] candidate pairs per row, d_offsets[N + 1] their exclusive prefix
// sums (d_offsets[N] = number of pairs); then the pairs (i, j, d[i][j] + d[j][i]) in the reference's loop order.
hipError_t launch_minmatch_scan(const float* d_matrix, int N, float threshold, float* d_min_values, int32_t* d_counts,
                                int64_t* d_offsets, hipStream_t st);
hipError_t launch_minmatch_fill(const float* d_matrix, int N, const float* d_min_values, const int64_t* d_offsets,
                                int64_t capacity, int32_t* d_ci, int32_t* d_cj, float* d_cd, hipStream_t st);

// Re-painting of one window from its stones (register-resident kernels, repaint_regs.hpp).
struct RepaintLaunch {
  const uint64_t* rows;
  int row_stride;
  const int32_t* dk;
  const SiteRec* rec;
  const int64_t* off;
  const int32_t* nd;
  const int32_t* bbj;
  const int32_t* bej;
  const int32_t* order;
  const uint8_t* stones;    // [K] records of window w
  const int64_t* row_off;   // [K] first row of each target
  const double* last_coef;  // [K]
  const double* last_nx;
  double* scratch;          // [repaint_checkpoint_rows(rows, K, MPL)][pitch] alpha checkpoints
  double* cp_sum;           // one double per checkpoint row
  float* topology;          // [rows][pitch]
  float* logscales;         // [rows]
  int w, pitch;
  int N, L, W, k_begin, K;
  PaintConsts c;
};
hipError_t launch_repaint(const RepaintLaunch& a, PaintGeometry g, hipStream_t st);
// Rows of a section that the re-painting walks between two alpha checkpoints: the backward launch keeps B - 1 alpha
// rows and the beta row in accumulation registers, 2 * MPL of them each, in what 256 registers per lane leave beside
// the kernel's 56 + 2 * MPL and the compiler's 32 AGPRs; at most 8.  1: every row is a checkpoint.
constexpr int repaint_block_rows(int MPL)
{
  const int slots = (256 - 32 - (56 + 2 * MPL)) / (2 * MPL);
  return slots < 2 ? 1 : (slots > 8 ? 8 : slots);
}
// Checkpoint rows of a window of `rows` topology rows over K targets: target t's start at row_off[t] / B + t.
constexpr long long repaint_checkpoint_rows(long long rows, int K, int MPL) { return rows / repaint_block_rows(MPL) + K + 1; }

}  // namespace relate
