// Host side of librelate_gpu.so: the C ABI of include/relate_gpu.h over the
// gfx950 kernels (paint_kernels.hip, paint_regs_g*.hip).  Mirrors, for the
// painting path only, what pipeline/Paint.cpp and src/data.cpp do on the host:
// reading the chunk files, the loop over targets, and the record files.
//
// There is no CPU fallback anywhere in this file: without a HIP device every
// compute entry point fails with RELATE_EGPU.
#include "../../include/relate_gpu.h"

#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>

#include "chunk_io.hpp"
#include "haps_loader.hpp"
#include "paint_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
  g_err = msg;
  return code;
}

}  // namespace

int relate::set_last_error(int code, const std::string& msg) { return fail(code, msg); }

namespace {

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(RELATE_EGPU, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Device buffer that grows and is reused between calls.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes)
  {
    if (bytes <= cap) return RELATE_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    // grow with some headroom so that successive batches of slightly different size do not reallocate
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) {
      p = nullptr;
      return fail(RELATE_ENOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed");
    }
    cap = want;
    return RELATE_OK;
  }
  void release()
  {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const
  {
    return static_cast<T*>(p);
  }
};

// Page-locked host buffer that grows and is reused between calls (small arrays that cross the link inside
// prepare: pageable memory would add a staging copy and a hidden wait to each transfer).
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes)
  {
    if (bytes <= cap) return RELATE_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
      p = nullptr;
      return fail(RELATE_ENOMEM, "hipHostMalloc of " + std::to_string(want) + " bytes failed");
    }
    cap = want;
    return RELATE_OK;
  }
  void release()
  {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const
  {
    return static_cast<T*>(p);
  }
};

relate::PaintConsts make_consts(int N, double theta_p, double theta_d)
{
  // src/fast_painting.hpp:19-45, with the host's libm as in the reference
  relate::PaintConsts c;
  const double ntheta = 1.0 - theta_p;
  c.Nm1 = (double)N - 1.0;
  const double prior_theta = theta_p / c.Nm1 - ntheta / c.Nm1;
  const double prior_ntheta = ntheta / c.Nm1;
  const double theta_ratio = theta_p / ntheta - 1.0;
  c.prior_mis = 1.0 * prior_theta + prior_ntheta;
  c.prior_match = prior_ntheta;
  c.rho_theta = 1.0 * theta_ratio + 1.0;
  c.log_ntheta = std::log(ntheta);
  c.log_small = std::log(0.01);
  c.log_Nm1 = std::log(c.Nm1);
  c.theta_d = theta_d;
  c.ntheta_d = 1.0 - theta_d;
  c.inv_theta_d = 1.0 / c.theta_d;
  c.inv_ntheta_d = 1.0 / c.ntheta_d;
  c.lo = relate::kRescaleLo;
  c.hi = 1.0 / c.lo;
  c.same_theta = theta_p == theta_d ? 1 : 0;
  c.reserved = 0;
  return c;
}

int check_windows(int L, int W, const int32_t* wb)
{
  if (W < 1 || !wb) return fail(RELATE_EINVAL, "need at least one window");
  if (wb[W] != L) return fail(RELATE_EINVAL, "last window boundary must equal L (src/fast_painting.cpp:42)");
  if (wb[0] < 0) return fail(RELATE_EINVAL, "negative window boundary");
  for (int w = 1; w <= W; w++) {
    if (wb[w] < wb[w - 1]) return fail(RELATE_EINVAL, "window boundaries must not decrease");
    if (wb[w] < 1) return fail(RELATE_EINVAL, "window boundaries after the first must be positive");
  }
  return RELATE_OK;
}

}  // namespace

struct relate_plan {
  int device = 0;
  hipStream_t own_stream = nullptr;
  int variant = 1, prefer_waves = 0;
  // chunk
  int N = 0, L = 0, W = 0;
  int row_stride = 0, col_stride = 0;
  DevBuf packed;  // rows, then cols
  bool packed_borrowed = false;
  const uint64_t* d_rows = nullptr;
  const uint64_t* d_cols = nullptr;
  DevBuf seq_own, r_own, wb_dev;
  const double* d_r = nullptr;
  std::vector<int32_t> wb;
  // prepared targets
  int k_begin = 0, K = 0;
  double theta_p = 0, theta_d = 0;
  DevBuf nd, off, dk, coef, nx, bbj, bej, order;
  DevBuf trace, rec;
  // re-painting
  std::vector<int32_t> h_bbj, h_bej;  // host copies of the stone indices, [K][W] (fetched when first asked for)
  HostBuf h_order;                    // launch order (int32); a member so that the asynchronous upload may outlive prepare
  bool stones_on_host = false;
  hipStream_t prepare_stream = nullptr;  // the null stream is a stream like any other: validity is a flag of its own
  bool have_prepare_stream = false;
  DevBuf row_off, last_coef, last_nx, scratch, cp_sum;
  DevBuf dm_args;  // per-row arguments of the distance-matrix kernel
  DevBuf mm_min, mm_counts, mm_off, mm_ci, mm_cj, mm_cd;  // MinMatch scans: row minima, pair counts / offsets, the pairs
  HostBuf h_mm;
  // page-locked staging of the small per-call arrays of repaint / distance_matrix: the upload is asynchronous, and
  // the next call waits for `staged` (the upload alone, not the kernels behind it) before it refills the block
  HostBuf h_stage;
  hipEvent_t staged = nullptr;
  bool stage_in_flight = false;
  HostBuf h_nd;   // int32 [K]
  HostBuf h_off;  // int64 [K]
  int64_t sum_nd = 0;
  relate::PaintGeometry geo{0, 0};
  relate::PaintGeometry geo_repaint{0, 0};  // re-painting is bound by memory latency: it takes more waves per target when there are few targets
  // timing
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool have_pack = false, have_prepare = false, have_paint = false;

  // A NULL `stream` argument means what it means in every HIP call: the default (null) stream, which is also
  // the handle torch reports for its default stream.  Work the library starts on its own behalf (host-buffer
  // and file entry points) runs on own_stream and is waited for before those calls return.
  hipStream_t pick(void* s) const { return static_cast<hipStream_t>(s); }
};

namespace {

int plan_set_chunk(relate_plan* p, int N, int L, int W, const int32_t* wb, hipStream_t st)
{
  if (N < 2 || L < 2) return fail(RELATE_EINVAL, "need N >= 2 haplotypes and L >= 2 SNPs");
  if (int rc = check_windows(L, W, wb)) return rc;
  p->N = N;
  p->L = L;
  p->W = W;
  p->row_stride = relate::row_stride_words(N);
  p->col_stride = relate::col_stride_words(L);
  p->wb.assign(wb, wb + W + 1);
  if (int rc = p->wb_dev.reserve(sizeof(int32_t) * (W + 1))) return rc;
  HIP_TRY(hipMemcpyAsync(p->wb_dev.p, p->wb.data(), sizeof(int32_t) * (W + 1), hipMemcpyHostToDevice, st));
  p->K = 0;
  p->have_prepare = false;
  return RELATE_OK;
}

// The staging block, at least `bytes` long, once the upload of the previous call has left it.
int stage_block(relate_plan* p, size_t bytes, void** out)
{
  if (p->stage_in_flight) {
    HIP_TRY(hipEventSynchronize(p->staged));
    p->stage_in_flight = false;
  }
  if (int rc = p->h_stage.reserve(bytes)) return rc;
  *out = p->h_stage.p;
  return RELATE_OK;
}

size_t rows_bytes(const relate_plan* p) { return relate::rows_alloc_bytes(p->N, p->L); }
size_t cols_bytes(const relate_plan* p) { return ((size_t)p->N * p->col_stride + 64) * sizeof(uint64_t); }

}  // namespace

extern "C" {

const char* relate_last_error(void) { return g_err.c_str(); }

int relate_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

size_t relate_record_bytes(int N) { return 56 + (size_t)8 * N; }

int relate_plan_create(int device, relate_plan** out)
{
  if (!out) return fail(RELATE_EINVAL, "null output pointer");
  *out = nullptr;
  int n = relate_device_count();
  if (n <= 0) return fail(RELATE_EGPU, "no HIP device visible: the painting path has no CPU fallback");
  if (device < 0 || device >= n) return fail(RELATE_EINVAL, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  relate_plan* p = new relate_plan();
  p->device = device;
  hipError_t e = hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking);
  for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&p->ev[i]);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->staged, hipEventDisableTiming);
  if (e != hipSuccess) {
    relate_plan_destroy(p);
    return fail(RELATE_EGPU, std::string("stream/event creation: ") + hipGetErrorString(e));
  }
  *out = p;
  return RELATE_OK;
}

void relate_plan_destroy(relate_plan* p)
{
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->own_stream) (void)hipStreamSynchronize(p->own_stream);
  if (!p->packed_borrowed) p->packed.release();
  for (DevBuf* b : {&p->seq_own, &p->r_own, &p->wb_dev, &p->nd, &p->off, &p->dk, &p->coef, &p->nx, &p->bbj, &p->bej,
                    &p->order, &p->trace, &p->rec, &p->row_off, &p->last_coef, &p->last_nx, &p->scratch, &p->dm_args,
                    &p->mm_min, &p->mm_counts, &p->mm_off, &p->mm_ci, &p->mm_cj, &p->mm_cd})
    b->release();
  for (HostBuf* b : {&p->h_nd, &p->h_off, &p->h_order, &p->h_stage, &p->h_mm}) b->release();
  if (p->staged) (void)hipEventDestroy(p->staged);
  for (auto& e : p->ev)
    if (e) (void)hipEventDestroy(e);
  if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
  delete p;
}

int relate_plan_configure(relate_plan* p, int variant, int waves_per_target)
{
  if (!p) return fail(RELATE_EINVAL, "null plan");
  if (variant != 0 && variant != 1) return fail(RELATE_EINVAL, "variant must be 0 or 1");
  if (waves_per_target != 0 && waves_per_target != 1 && waves_per_target != 2 && waves_per_target != 4 &&
      waves_per_target != 8)
    return fail(RELATE_EINVAL, "waves_per_target must be 0, 1, 2, 4 or 8");
  p->variant = variant;
  p->prefer_waves = waves_per_target;
  p->K = 0;  // what prepare leaves behind depends on the variant: prepare again before the next paint
  p->have_prepare = false;
  return RELATE_OK;
}

size_t relate_packed_bytes(int N, int L)
{
  if (N < 2 || L < 2) return 0;
  return relate::rows_alloc_bytes(N, L) + ((size_t)N * relate::col_stride_words(L) + 64) * sizeof(uint64_t);
}

static int load_device_impl(relate_plan* p, int N, int L, const void* d_seq, const void* d_r, int W, const int32_t* wb,
                            void* d_packed, size_t packed_bytes, void* stream)
{
  if (!p || !d_seq || !d_r) return fail(RELATE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = p->pick(stream);
  if (int rc = plan_set_chunk(p, N, L, W, wb, st)) return rc;
  const size_t rb = rows_bytes(p), cb = cols_bytes(p);
  if (d_packed) {
    if (packed_bytes != rb + cb) return fail(RELATE_EINVAL, "packed buffer has the wrong size for N and L");
    if (reinterpret_cast<uintptr_t>(d_packed) % 256 != 0) return fail(RELATE_EINVAL, "packed buffer must be 256-byte aligned");
    if (!p->packed_borrowed) p->packed.release();
    p->packed.p = d_packed;
    p->packed.cap = packed_bytes;
    p->packed_borrowed = true;
  } else {
    if (p->packed_borrowed) {
      p->packed = DevBuf();
      p->packed_borrowed = false;
    }
    if (int rc = p->packed.reserve(rb + cb)) return rc;
  }
  uint64_t* rows = p->packed.as<uint64_t>();
  uint64_t* cols = reinterpret_cast<uint64_t*>(static_cast<char*>(p->packed.p) + rb);
  HIP_TRY(hipEventRecord(p->ev[0], st));
  // slack after the last row is read (never used) by the recursion kernel: keep it defined
  HIP_TRY(hipMemsetAsync(reinterpret_cast<char*>(rows) + (size_t)L * p->row_stride * 8, 0,
                         rb - (size_t)L * p->row_stride * 8, st));
  HIP_TRY(relate::launch_pack_rows(static_cast<const uint8_t*>(d_seq), rows, N, L, p->row_stride, st));
  HIP_TRY(relate::launch_transpose(rows, cols, N, L, p->row_stride, p->col_stride, st));
  HIP_TRY(hipEventRecord(p->ev[1], st));
  p->have_pack = true;
  p->d_rows = rows;
  p->d_cols = cols;
  p->d_r = static_cast<const double*>(d_r);
  return RELATE_OK;
}

int relate_plan_load_device(relate_plan* p, int N, int L, const void* d_seq, const void* d_r, int W,
                            const int32_t* wb, void* stream)
{
  return load_device_impl(p, N, L, d_seq, d_r, W, wb, nullptr, 0, stream);
}

int relate_plan_load_device_into(relate_plan* p, int N, int L, const void* d_seq, const void* d_r, int W,
                                 const int32_t* wb, void* d_packed, size_t packed_bytes, void* stream)
{
  if (!d_packed) return fail(RELATE_EINVAL, "null packed buffer");
  return load_device_impl(p, N, L, d_seq, d_r, W, wb, d_packed, packed_bytes, stream);
}

int relate_plan_load_host(relate_plan* p, int N, int L, const uint8_t* seq, const double* r, int W, const int32_t* wb)
{
  if (!p || !seq || !r) return fail(RELATE_EINVAL, "null argument");
  if (N < 2 || L < 2) return fail(RELATE_EINVAL, "need N >= 2 haplotypes and L >= 2 SNPs");
  HIP_TRY(hipSetDevice(p->device));
  if (int rc = p->seq_own.reserve((size_t)N * L)) return rc;
  if (int rc = p->r_own.reserve(sizeof(double) * (size_t)L)) return rc;
  HIP_TRY(hipMemcpyAsync(p->seq_own.p, seq, (size_t)N * L, hipMemcpyHostToDevice, p->own_stream));
  HIP_TRY(hipMemcpyAsync(p->r_own.p, r, sizeof(double) * (size_t)L, hipMemcpyHostToDevice, p->own_stream));
  int rc = relate_plan_load_device(p, N, L, p->seq_own.p, p->r_own.p, W, wb, p->own_stream);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(p->own_stream));
  p->seq_own.release();  // the byte matrix is 8x the packed one: drop it once packed
  return RELATE_OK;
}

int relate_row_stride_words(int N) { return N < 2 ? 0 : relate::row_stride_words(N); }

int relate_plan_load_rows_host(relate_plan* p, int N, int L, const uint64_t* rows, const double* r, double rho, int W,
                               const int32_t* wb)
{
  if (!p || !rows || !r) return fail(RELATE_EINVAL, "null argument");
  if (N < 2 || L < 2) return fail(RELATE_EINVAL, "need N >= 2 haplotypes and L >= 2 SNPs");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = p->own_stream;
  if (int rc = plan_set_chunk(p, N, L, W, wb, st)) return rc;
  const size_t rb = rows_bytes(p), cb = cols_bytes(p), used = (size_t)L * p->row_stride * sizeof(uint64_t);
  if (p->packed_borrowed) {
    p->packed = DevBuf();
    p->packed_borrowed = false;
  }
  if (int rc = p->packed.reserve(rb + cb)) return rc;
  if (int rc = p->r_own.reserve(sizeof(double) * (size_t)L)) return rc;
  uint64_t* d_rows = p->packed.as<uint64_t>();
  uint64_t* d_cols = reinterpret_cast<uint64_t*>(static_cast<char*>(p->packed.p) + rb);
  // r times rho (Data::SetPainting, src/data.cpp:124-129), in page-locked staging so that the upload is a plain DMA
  void* stage = nullptr;
  if (int rc = stage_block(p, sizeof(double) * (size_t)L, &stage)) return rc;
  double* hr = static_cast<double*>(stage);
  for (int s = 0; s < L; s++) hr[s] = r[s] * rho;
  HIP_TRY(hipEventRecord(p->ev[0], st));
  HIP_TRY(hipMemcpyAsync(p->r_own.p, hr, sizeof(double) * (size_t)L, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_rows, rows, used, hipMemcpyHostToDevice, st));
  // slack after the last row is read (never used) by the recursion kernel: keep it defined
  HIP_TRY(hipMemsetAsync(reinterpret_cast<char*>(d_rows) + used, 0, rb - used, st));
  HIP_TRY(relate::launch_transpose(d_rows, d_cols, N, L, p->row_stride, p->col_stride, st));
  HIP_TRY(hipEventRecord(p->ev[1], st));
  HIP_TRY(hipStreamSynchronize(st));  // the caller's rows and the staging block are free again
  p->have_pack = true;
  p->d_rows = d_rows;
  p->d_cols = d_cols;
  p->d_r = p->r_own.as<double>();
  return RELATE_OK;
}

int relate_plan_load_haps_chunk(relate_plan* p, const relate_haps* h, int chunk, double rho)
{
  if (!p || !h) return fail(RELATE_EINVAL, "null argument");
  if (chunk < 0 || chunk >= (int)h->chunks.size()) return fail(RELATE_EINVAL, "no such chunk");
  const relate::HapsChunk& c = h->chunks[(size_t)chunk];
  if (h->row_stride != relate::row_stride_words(h->N)) return fail(RELATE_EINVAL, "data set packed with another row width");
  if ((size_t)(c.first_snp + c.L) * h->row_stride > h->rows.size()) return fail(RELATE_EINVAL, "chunk lies outside the data set");
  return relate_plan_load_rows_host(p, h->N, c.L, h->rows.data() + (size_t)c.first_snp * h->row_stride,
                                    h->r.data() + c.first_snp, rho, (int)c.wb.size() - 1, c.wb.data());
}

int relate_plan_packed(relate_plan* p, void** d_packed, size_t* bytes)
{
  if (!p || !d_packed || !bytes) return fail(RELATE_EINVAL, "null argument");
  if (!p->d_rows) return fail(RELATE_EINVAL, "no chunk loaded");
  *d_packed = const_cast<uint64_t*>(p->d_rows);
  *bytes = rows_bytes(p) + cols_bytes(p);
  return RELATE_OK;
}

int relate_plan_load_packed(relate_plan* p, int N, int L, const void* d_packed, size_t bytes, const void* d_r, int W,
                            const int32_t* wb, void* stream)
{
  if (!p || !d_packed || !d_r) return fail(RELATE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = p->pick(stream);
  if (int rc = plan_set_chunk(p, N, L, W, wb, st)) return rc;
  const size_t rb = rows_bytes(p), cb = cols_bytes(p);
  if (bytes != rb + cb) return fail(RELATE_EINVAL, "packed matrix has the wrong size for N and L");
  if (!p->packed_borrowed) p->packed.release();
  p->packed_borrowed = true;
  p->packed.p = const_cast<void*>(d_packed);
  p->packed.cap = bytes;
  p->d_rows = static_cast<const uint64_t*>(d_packed);
  p->d_cols = reinterpret_cast<const uint64_t*>(static_cast<const char*>(d_packed) + rb);
  p->d_r = static_cast<const double*>(d_r);
  p->have_pack = false;
  return RELATE_OK;
}

int relate_plan_prepare(relate_plan* p, double theta_p, double theta_d, int k_begin, int k_end, int64_t* sum_derived,
                        void* stream)
{
  if (!p) return fail(RELATE_EINVAL, "null plan");
  if (!p->d_rows) return fail(RELATE_EINVAL, "no chunk loaded");
  if (k_begin < 0 || k_end > p->N || k_begin >= k_end) return fail(RELATE_EINVAL, "bad target range");
  if (!(theta_p > 0.0 && theta_p < 1.0 && theta_d > 0.0 && theta_d < 1.0))
    return fail(RELATE_EINVAL, "theta must lie in (0,1) (src/fast_painting.hpp:23,37)");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = p->pick(stream);
  const int K = k_end - k_begin, W = p->W;
  p->geo = relate::choose_geometry_for_launch(p->N, K, p->prefer_waves, p->variant);
  if (p->geo.MPL <= 0) return fail(RELATE_ERANGE, "N is too large for the selected kernel variant");
  p->geo_repaint = relate::choose_geometry(p->N, p->prefer_waves, p->variant);  // one window per launch: its own rule below
  if (p->variant >= 1 && p->prefer_waves == 0) {
    // one window of K targets should fill most of the 4096 wave slots of the chip (4 per SIMD); measured on
    // C2 (K = 2000): 2 waves per target 4.8 ms, 1 wave 5.9 ms, 4 waves 5.3 ms
    while (p->geo_repaint.G < 8 && (long long)K * p->geo_repaint.G < 3072) {
      const relate::PaintGeometry g2 = relate::choose_geometry(p->N, 2 * p->geo_repaint.G, 1);
      if (g2.G != 2 * p->geo_repaint.G || g2.MPL <= 0) break;
      p->geo_repaint = g2;
    }
  }
  const relate::PaintConsts c = make_consts(p->N, theta_p, theta_d);

  HIP_TRY(hipEventRecord(p->ev[2], st));
  if (int rc = p->nd.reserve(sizeof(int32_t) * (size_t)K)) return rc;
  HIP_TRY(relate::launch_count_derived(p->d_cols, p->col_stride, p->L, k_begin, K, p->nd.as<int32_t>(), st));
  // the page-locked arrays below may still be feeding the uploads of an earlier prepare on another stream
  if (p->have_prepare_stream && p->prepare_stream != st) HIP_TRY(hipStreamSynchronize(p->prepare_stream));
  if (int rc = p->h_nd.reserve(sizeof(int32_t) * (size_t)K)) return rc;
  if (int rc = p->h_off.reserve(sizeof(int64_t) * (size_t)K)) return rc;
  if (int rc = p->h_order.reserve(sizeof(int32_t) * (size_t)K)) return rc;
  int32_t* h_nd = p->h_nd.as<int32_t>();
  int64_t* h_off = p->h_off.as<int64_t>();
  int32_t* order = p->h_order.as<int32_t>();
  HIP_TRY(hipMemcpyAsync(h_nd, p->nd.p, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int64_t total = 0;
  int max_nd = 0;
  for (int t = 0; t < K; t++) {
    h_off[t] = total;
    total += h_nd[t];
    max_nd = std::max(max_nd, h_nd[t]);
    if (h_nd[t] < 2 || h_nd[t] > p->L) return fail(RELATE_EGPU, "derived-site count out of range (kernel fault?)");
  }
  p->sum_nd = total;
  // longest targets first: workgroups are handed out in launch order, so the tail of the launch is short
  std::iota(order, order + K, 0);
  std::stable_sort(order, order + K, [&](int a, int b) { return h_nd[a] > h_nd[b]; });

  if (int rc = p->off.reserve(sizeof(int64_t) * (size_t)K)) return rc;
  if (int rc = p->order.reserve(sizeof(int32_t) * (size_t)K)) return rc;
  if (int rc = p->dk.reserve(sizeof(int32_t) * (size_t)total)) return rc;
  const bool records = p->variant >= 1;  // the register-resident kernels read interleaved site records
  if (records) {
    if (int rc = p->rec.reserve(sizeof(relate::SiteRec) * (size_t)total)) return rc;
  } else {
    if (int rc = p->coef.reserve(sizeof(double) * (size_t)total)) return rc;
    if (int rc = p->nx.reserve(sizeof(double) * (size_t)total)) return rc;
  }
  if (int rc = p->bbj.reserve(sizeof(int32_t) * (size_t)K * W)) return rc;
  if (int rc = p->bej.reserve(sizeof(int32_t) * (size_t)K * W)) return rc;
  HIP_TRY(hipMemcpyAsync(p->off.p, h_off, sizeof(int64_t) * (size_t)K, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(p->order.p, order, sizeof(int32_t) * (size_t)K, hipMemcpyHostToDevice, st));
  HIP_TRY(relate::launch_fill_sites(p->d_cols, p->col_stride, p->L, k_begin, K, p->off.as<int64_t>(),
                                    p->dk.as<int32_t>(), st));
  HIP_TRY(relate::launch_site_terms(p->dk.as<int32_t>(), p->off.as<int64_t>(), p->nd.as<int32_t>(), K, max_nd, p->d_r,
                                    p->L, c, records ? nullptr : p->coef.as<double>(), records ? nullptr : p->nx.as<double>(),
                                    records ? p->rec.as<relate::SiteRec>() : nullptr, st));
  HIP_TRY(relate::launch_window_stones(p->dk.as<int32_t>(), p->off.as<int64_t>(), p->nd.as<int32_t>(), K, W,
                                       p->wb_dev.as<int32_t>(), p->bbj.as<int32_t>(), p->bej.as<int32_t>(), st));
  HIP_TRY(hipEventRecord(p->ev[3], st));
  // No wait here: the recursion kernel can follow on the same stream at once.  The host copies of the stone
  // indices are fetched by the first call that needs them (stone_indices_on_host).
  p->stones_on_host = false;
  p->prepare_stream = st;
  p->have_prepare_stream = true;
  p->have_prepare = true;
  p->k_begin = k_begin;
  p->K = K;
  p->theta_p = theta_p;
  p->theta_d = theta_d;
  if (sum_derived) *sum_derived = total;
  return RELATE_OK;
}

int relate_plan_paint(relate_plan* p, void* d_out, void* stream)
{
  if (!p || !d_out) return fail(RELATE_EINVAL, "null argument");
  if (p->K <= 0) return fail(RELATE_EINVAL, "no targets prepared");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = p->pick(stream);
  relate::PaintArgs a;
  a.rows = p->d_rows;
  a.row_stride = p->row_stride;
  a.dk = p->dk.as<int32_t>();
  a.coef = p->coef.as<double>();
  a.nx = p->nx.as<double>();
  a.rec = p->rec.as<relate::SiteRec>();
  a.off = p->off.as<int64_t>();
  a.nd = p->nd.as<int32_t>();
  a.bbj = p->bbj.as<int32_t>();
  a.bej = p->bej.as<int32_t>();
  a.order = p->order.as<int32_t>();
  a.wb = p->wb_dev.as<int32_t>();
  a.out = static_cast<uint8_t*>(d_out);
  a.N = p->N;
  a.L = p->L;
  a.W = p->W;
  a.k_begin = p->k_begin;
  a.K = p->K;
  a.c = make_consts(p->N, p->theta_p, p->theta_d);
  a.trace = nullptr;
  // Diagnostics: RELATE_TRACE=<file> makes the register-resident kernel record, per workgroup, where it
  // ran and when it started and ended (4 x u64: hardware id, start, end in 100 MHz ticks, pass | nd << 8).
  const char* trace_path = p->variant >= 1 ? std::getenv("RELATE_TRACE") : nullptr;
  const size_t trace_words = (size_t)8 * p->K;
  if (trace_path && *trace_path) {
    if (int rc = p->trace.reserve(sizeof(uint64_t) * trace_words)) return rc;
    a.trace = p->trace.as<uint64_t>();
  }
  HIP_TRY(hipEventRecord(p->ev[4], st));
  HIP_TRY(relate::launch_paint(a, p->geo, p->variant, st));
  HIP_TRY(hipEventRecord(p->ev[5], st));
  p->have_paint = true;
  if (a.trace) {
    std::vector<uint64_t> h(trace_words);
    HIP_TRY(hipMemcpyAsync(h.data(), a.trace, sizeof(uint64_t) * trace_words, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (FILE* f = std::fopen(trace_path, "wb")) {
      std::fwrite(h.data(), sizeof(uint64_t), trace_words, f);
      std::fclose(f);
    }
  }
  return RELATE_OK;
}

static int stone_indices_on_host(relate_plan* p)
{
  if (p->stones_on_host) return RELATE_OK;
  HIP_TRY(hipSetDevice(p->device));
  const size_t n = (size_t)p->K * p->W;
  p->h_bbj.resize(n);
  p->h_bej.resize(n);
  HIP_TRY(hipMemcpyAsync(p->h_bbj.data(), p->bbj.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, p->prepare_stream));
  HIP_TRY(hipMemcpyAsync(p->h_bej.data(), p->bej.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, p->prepare_stream));
  HIP_TRY(hipStreamSynchronize(p->prepare_stream));
  p->stones_on_host = true;
  return RELATE_OK;
}

int relate_plan_repaint_layout(relate_plan* p, int window, int64_t* total_rows, int* pitch, int64_t* row_offsets)
{
  if (!p) return fail(RELATE_EINVAL, "null plan");
  if (p->K <= 0 || !p->have_prepare) return fail(RELATE_EINVAL, "no targets prepared");
  if (int rc = stone_indices_on_host(p)) return rc;
  if (p->variant != 1) return fail(RELATE_EINVAL, "re-painting needs the register-resident kernel (variant 1)");
  if (window < 0 || window >= p->W) return fail(RELATE_EINVAL, "window out of range");
  int64_t rows = 0;
  for (int t = 0; t < p->K; t++) {
    if (row_offsets) row_offsets[t] = rows;
    const int n = p->h_bej[(size_t)t * p->W + window] - p->h_bbj[(size_t)t * p->W + window] + 1;
    if (n < 2) return fail(RELATE_EGPU, "section with fewer than two sites (kernel fault?)");
    rows += n;
  }
  if (row_offsets) row_offsets[p->K] = rows;
  if (total_rows) *total_rows = rows;
  if (pitch) *pitch = p->geo_repaint.G * p->geo_repaint.MPL * 64;
  return RELATE_OK;
}

int relate_plan_repaint(relate_plan* p, int window, const void* d_stones, void* d_topology, void* d_logscales,
                        void* stream)
{
  if (!p || !d_stones || !d_topology || !d_logscales) return fail(RELATE_EINVAL, "null argument");
  if (!p->have_prepare || p->K <= 0) return fail(RELATE_EINVAL, "no targets prepared");
  HIP_TRY(hipSetDevice(p->device));
  void* blk = nullptr;
  if (int rc = stage_block(p, sizeof(int64_t) * ((size_t)p->K + 1), &blk)) return rc;
  int64_t* ro = static_cast<int64_t*>(blk);
  int64_t rows = 0;
  int pitch = 0;
  if (int rc = relate_plan_repaint_layout(p, window, &rows, &pitch, ro)) return rc;
  hipStream_t st = p->pick(stream);
  const int K = p->K;
  if (int rc = p->row_off.reserve(sizeof(int64_t) * (size_t)K)) return rc;
  if (int rc = p->last_coef.reserve(sizeof(double) * (size_t)K)) return rc;
  if (int rc = p->last_nx.reserve(sizeof(double) * (size_t)K)) return rc;
  const size_t cp_rows = (size_t)relate::repaint_checkpoint_rows(rows, K, p->geo_repaint.MPL);
  if (int rc = p->scratch.reserve(sizeof(double) * cp_rows * pitch)) return rc;
  if (int rc = p->cp_sum.reserve(sizeof(double) * cp_rows)) return rc;
  HIP_TRY(hipMemcpyAsync(p->row_off.p, ro, sizeof(int64_t) * (size_t)K, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(p->staged, st));
  p->stage_in_flight = true;
  const relate::PaintConsts c = make_consts(p->N, p->theta_p, p->theta_d);
  HIP_TRY(relate::launch_section_last_terms(p->dk.as<int32_t>(), p->off.as<int64_t>(), p->bej.as<int32_t>(), K, p->W,
                                            window, p->d_r, c, p->last_coef.as<double>(), p->last_nx.as<double>(), st));
  relate::RepaintLaunch a;
  a.rows = p->d_rows;
  a.row_stride = p->row_stride;
  a.dk = p->dk.as<int32_t>();
  a.rec = p->rec.as<relate::SiteRec>();
  a.off = p->off.as<int64_t>();
  a.nd = p->nd.as<int32_t>();
  a.bbj = p->bbj.as<int32_t>();
  a.bej = p->bej.as<int32_t>();
  a.order = p->order.as<int32_t>();
  a.stones = static_cast<const uint8_t*>(d_stones);
  a.row_off = p->row_off.as<int64_t>();
  a.last_coef = p->last_coef.as<double>();
  a.last_nx = p->last_nx.as<double>();
  a.scratch = p->scratch.as<double>();
  a.cp_sum = p->cp_sum.as<double>();
  a.topology = static_cast<float*>(d_topology);
  a.logscales = static_cast<float*>(d_logscales);
  a.w = window;
  a.pitch = pitch;
  a.N = p->N;
  a.L = p->L;
  a.W = p->W;
  a.k_begin = p->k_begin;
  a.K = K;
  a.c = c;
  HIP_TRY(hipEventRecord(p->ev[4], st));
  HIP_TRY(relate::launch_repaint(a, p->geo_repaint, st));
  HIP_TRY(hipEventRecord(p->ev[5], st));
  p->have_paint = true;
  return RELATE_OK;
}

}  // extern "C"

namespace {

// Per-row arguments of the distance-matrix kernel as one block, laid out the same way on the host and on the
// device: row indices, the two weights, the two exponentials, the gathered pair of logscales, the flags.
struct DmBlock {
  size_t row, wl, wr, epn, enp, pair, flag, bytes;
  explicit DmBlock(int N)
      : row(0), wl((size_t)8 * N), wr((size_t)16 * N), epn((size_t)24 * N), enp((size_t)28 * N), pair((size_t)32 * N),
        flag((size_t)40 * N), bytes((size_t)41 * N)
  {
  }
};

int check_distance_rows(int N, int64_t total_rows, const int64_t* row_index, const uint8_t* interpolate,
                        const double* weight_left, const double* weight_right)
{
  bool any_mixed = false;
  for (int n = 0; n < N; n++) {
    any_mixed = any_mixed || interpolate[n];
    if (row_index[n] < 0 || row_index[n] + (interpolate[n] ? 1 : 0) >= total_rows)
      return fail(RELATE_EINVAL, "topology row index out of range");
  }
  if (any_mixed && (!weight_left || !weight_right)) return fail(RELATE_EINVAL, "interpolated rows need weights");
  return RELATE_OK;
}

// float exp_logscale_prev_next = exp(logscale_prev - logscale_next), and the other way round
// (src/anc_builder.cpp:171-172): a float difference, the C library's exp on it as a double, narrowed.
inline void logscale_exponentials(float ls_prev, float ls_next, float* e_pn, float* e_np)
{
  const float d_pn = ls_prev - ls_next, d_np = ls_next - ls_prev;
  *e_pn = (float)std::exp((double)d_pn);
  *e_np = (float)std::exp((double)d_np);
}

// Fills the host block h, uploads it into the device block d and launches the kernel on `st`.
// h_ls: a host copy of the logscales (total_rows floats), or NULL: then the 2N values needed are gathered on
// the device and fetched, which waits for `st` once.  h must stay untouched until the upload has left it
// (`uploaded`, if given, is recorded behind the last upload; without it the function waits itself).
int distance_matrix_run(int N, const float* d_top, int pitch, const float* d_ls, const float* h_ls,
                        const int64_t* row_index, const uint8_t* interpolate, const double* weight_left,
                        const double* weight_right, uint8_t* h, uint8_t* d, float* d_matrix, hipStream_t st,
                        hipEvent_t uploaded)
{
  const DmBlock b(N);
  int64_t* h_row = reinterpret_cast<int64_t*>(h + b.row);
  double* h_wl = reinterpret_cast<double*>(h + b.wl);
  double* h_wr = reinterpret_cast<double*>(h + b.wr);
  float* h_epn = reinterpret_cast<float*>(h + b.epn);
  float* h_enp = reinterpret_cast<float*>(h + b.enp);
  float* h_pair = reinterpret_cast<float*>(h + b.pair);
  uint8_t* h_flag = h + b.flag;
  for (int n = 0; n < N; n++) {
    h_row[n] = row_index[n];
    h_wl[n] = interpolate[n] ? weight_left[n] : 0.0;
    h_wr[n] = interpolate[n] ? weight_right[n] : 0.0;
    h_flag[n] = interpolate[n] ? 1 : 0;
  }
  if (h_ls) {
    for (int n = 0; n < N; n++) {
      h_pair[2 * n] = h_ls[row_index[n]];
      h_pair[2 * n + 1] = h_ls[row_index[n] + (interpolate[n] ? 1 : 0)];
    }
  } else {
    HIP_TRY(hipMemcpyAsync(d + b.row, h + b.row, (size_t)8 * N, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + b.flag, h + b.flag, (size_t)N, hipMemcpyHostToDevice, st));
    HIP_TRY(relate::launch_gather_logscales(d_ls, reinterpret_cast<const int64_t*>(d + b.row), d + b.flag, N,
                                            reinterpret_cast<float*>(d + b.pair), st));
    HIP_TRY(hipMemcpyAsync(h + b.pair, d + b.pair, (size_t)8 * N, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  for (int n = 0; n < N; n++) {
    if (interpolate[n]) logscale_exponentials(h_pair[2 * n], h_pair[2 * n + 1], &h_epn[n], &h_enp[n]);
    else h_epn[n] = h_enp[n] = 1.0f;
  }
  if (h_ls) HIP_TRY(hipMemcpyAsync(d, h, b.bytes, hipMemcpyHostToDevice, st));
  else HIP_TRY(hipMemcpyAsync(d + b.wl, h + b.wl, b.pair - b.wl, hipMemcpyHostToDevice, st));
  if (uploaded) HIP_TRY(hipEventRecord(uploaded, st));
  HIP_TRY(relate::launch_distance_matrix(d_top, pitch, d_ls, reinterpret_cast<const int64_t*>(d + b.row), d + b.flag,
                                         reinterpret_cast<const double*>(d + b.wl), reinterpret_cast<const double*>(d + b.wr),
                                         reinterpret_cast<const float*>(d + b.epn), reinterpret_cast<const float*>(d + b.enp),
                                         N, d_matrix, st));
  if (!uploaded) HIP_TRY(hipStreamSynchronize(st));
  return RELATE_OK;
}

// relate_plan_distance_matrix with an optional host copy of the logscales (the library's DistanceMeasure keeps one)
int plan_distance_matrix(relate_plan* p, int N, const void* d_topology, int pitch, const void* d_logscales,
                         const float* h_logscales, int64_t total_rows, const int64_t* row_index, const uint8_t* interpolate,
                         const double* weight_left, const double* weight_right, void* d_matrix, void* stream)
{
  if (!p || !d_topology || !d_logscales || !row_index || !interpolate || !d_matrix)
    return fail(RELATE_EINVAL, "null argument");
  if (N < 1 || pitch < N) return fail(RELATE_EINVAL, "bad matrix size or pitch");
  if (int rc = check_distance_rows(N, total_rows, row_index, interpolate, weight_left, weight_right)) return rc;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = p->pick(stream);
  const DmBlock b(N);
  void* blk = nullptr;
  if (int rc = stage_block(p, b.bytes, &blk)) return rc;
  if (int rc = p->dm_args.reserve(b.bytes)) return rc;
  HIP_TRY(hipEventRecord(p->ev[4], st));
  p->stage_in_flight = true;  // from here on the block may be feeding an upload
  if (int rc = distance_matrix_run(N, static_cast<const float*>(d_topology), pitch, static_cast<const float*>(d_logscales),
                                   h_logscales, row_index, interpolate, weight_left, weight_right, static_cast<uint8_t*>(blk),
                                   p->dm_args.as<uint8_t>(), static_cast<float*>(d_matrix), st, p->staged))
    return rc;
  HIP_TRY(hipEventRecord(p->ev[5], st));
  p->have_paint = true;
  return RELATE_OK;
}

}  // namespace

extern "C" {

int relate_plan_distance_matrix(relate_plan* p, int N, const void* d_topology, int pitch, const void* d_logscales,
                                int64_t total_rows, const int64_t* row_index, const uint8_t* interpolate,
                                const double* weight_left, const double* weight_right, void* d_matrix, void* stream)
{
  return plan_distance_matrix(p, N, d_topology, pitch, d_logscales, nullptr, total_rows, row_index, interpolate, weight_left,
                              weight_right, d_matrix, stream);
}

// --------------------------------------------------------------------------
// The distance matrix for callers that cannot allocate (tight loops that ask for a matrix per tree, harnesses
// that forbid hipMalloc).  relate_plan_distance_matrix stages its per-row arguments in a device buffer and a
// page-locked host block that it allocates and grows; here they go into a fixed array in the code object, one
// copy per GPU, of which the host keeps the address per device, and a fixed host array of the same size.
// --------------------------------------------------------------------------
}  // extern "C"

namespace {

constexpr int kNoAllocMaxN = 36864;  // the widest panel the painting kernels take (8 waves x 72 words)
constexpr size_t kNoAllocBytes = ((size_t)kNoAllocMaxN * 41 + 255) / 256 * 256;  // one DmBlock of the widest panel
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];
alignas(64) unsigned char g_noalloc_host[kNoAllocBytes];  // host side of the same block; one call at a time (mutex)
std::mutex g_noalloc_host_mutex;
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};

// Address of this device's copy of `scratch` (the device must be current).  In host code the array's name is
// only a handle for the symbol, never an address a kernel can use.
int scratch_address(int device, unsigned char** out)
{
  std::lock_guard<std::mutex> lk(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) return fail(RELATE_EGPU, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    g_scratch_ptr[device] = ptr;
  }
  *out = static_cast<unsigned char*>(g_scratch_ptr[device]);
  return RELATE_OK;
}

}  // namespace

extern "C" {

int relate_distance_matrix_noalloc(int device, int N, const void* d_topology, int pitch, const void* d_logscales,
                                   int64_t total_rows, const int64_t* row_index, const uint8_t* interpolate,
                                   const double* weight_left, const double* weight_right, void* d_matrix, void* stream)
{
  if (!d_topology || !d_logscales || !row_index || !interpolate || !d_matrix) return fail(RELATE_EINVAL, "null argument");
  if (N < 1 || pitch < N) return fail(RELATE_EINVAL, "bad matrix size or pitch");
  if (N > kNoAllocMaxN)
    return fail(RELATE_ERANGE, "relate_distance_matrix_noalloc takes at most " + std::to_string(kNoAllocMaxN) +
                                   " rows (its argument array is fixed): use relate_plan_distance_matrix");
  const int ndev = relate_device_count();
  if (ndev <= 0) return fail(RELATE_EGPU, "no HIP device visible");
  if (device < 0 || device >= ndev || device >= kMaxDevices) return fail(RELATE_EINVAL, "device index out of range");
  if (int rc = check_distance_rows(N, total_rows, row_index, interpolate, weight_left, weight_right)) return rc;
  HIP_TRY(hipSetDevice(device));
  unsigned char* d = nullptr;
  if (int rc = scratch_address(device, &d)) return rc;
  // the host block is shared by every call in the process: held until the uploads have left it (the call
  // returns after the stream has taken them, see distance_matrix_run without an event)
  std::lock_guard<std::mutex> lk(g_noalloc_host_mutex);
  return distance_matrix_run(N, static_cast<const float*>(d_topology), pitch, static_cast<const float*>(d_logscales), nullptr,
                             row_index, interpolate, weight_left, weight_right, g_noalloc_host, d,
                             static_cast<float*>(d_matrix), static_cast<hipStream_t>(stream), nullptr);
}

float relate_minmatch_threshold(double theta) { return (float)(-0.2 * std::log(theta / (1.0 - theta))); }

int relate_plan_minmatch_scan(relate_plan* p, int N, const void* d_matrix, float threshold, float* min_values,
                              int64_t capacity, int32_t* cand_i, int32_t* cand_j, float* cand_dist, int64_t* count,
                              void* stream)
{
  if (!p || !d_matrix || !min_values || !count) return fail(RELATE_EINVAL, "null argument");
  if (N < 1 || capacity < 0) return fail(RELATE_EINVAL, "bad matrix size or capacity");
  if (capacity > 0 && (!cand_i || !cand_j || !cand_dist)) return fail(RELATE_EINVAL, "null candidate arrays");
  *count = 0;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = p->pick(stream);
  if (int rc = p->mm_min.reserve(sizeof(float) * (size_t)N)) return rc;
  if (int rc = p->mm_counts.reserve(sizeof(int32_t) * (size_t)N)) return rc;
  if (int rc = p->mm_off.reserve(sizeof(int64_t) * ((size_t)N + 1))) return rc;
  HIP_TRY(hipEventRecord(p->ev[4], st));
  HIP_TRY(relate::launch_minmatch_scan(static_cast<const float*>(d_matrix), N, threshold, p->mm_min.as<float>(),
                                       p->mm_counts.as<int32_t>(), p->mm_off.as<int64_t>(), st));
  // page-locked landing block: the total first, later the pairs
  if (int rc = p->h_mm.reserve(sizeof(int64_t) + sizeof(float) * (size_t)N)) return rc;
  int64_t* h_total = p->h_mm.as<int64_t>();
  HIP_TRY(hipMemcpyAsync(h_total, p->mm_off.as<int64_t>() + N, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(min_values, p->mm_min.p, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int64_t total = *h_total;
  if (total < 0 || total > (int64_t)N * (N - 1) / 2) return fail(RELATE_EGPU, "candidate count out of range (kernel fault?)");
  *count = total;
  if (total > capacity)
    return fail(RELATE_ERANGE, "the matrix has " + std::to_string(total) + " candidate pairs, the arrays hold " +
                                   std::to_string(capacity) + ": call again with larger ones");
  if (total > 0) {
    if (int rc = p->mm_ci.reserve(sizeof(int32_t) * (size_t)total)) return rc;
    if (int rc = p->mm_cj.reserve(sizeof(int32_t) * (size_t)total)) return rc;
    if (int rc = p->mm_cd.reserve(sizeof(float) * (size_t)total)) return rc;
    HIP_TRY(relate::launch_minmatch_fill(static_cast<const float*>(d_matrix), N, p->mm_min.as<float>(), p->mm_off.as<int64_t>(),
                                         total, p->mm_ci.as<int32_t>(), p->mm_cj.as<int32_t>(), p->mm_cd.as<float>(), st));
    HIP_TRY(hipMemcpyAsync(cand_i, p->mm_ci.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(cand_j, p->mm_cj.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(cand_dist, p->mm_cd.p, sizeof(float) * (size_t)total, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipEventRecord(p->ev[5], st));
  p->have_paint = true;
  HIP_TRY(hipStreamSynchronize(st));
  return RELATE_OK;
}

int relate_minmatch_apply_candidates(int N, int64_t count, const int32_t* cand_i, const int32_t* cand_j,
                                     const float* cand_dist, const double* sample_ages, double age,
                                     double (*draw)(void*), void* draw_ctx, relate_candidate* mcandidates,
                                     relate_candidate* best)
{
  if (N < 1 || count < 0 || !draw || !mcandidates || !best) return fail(RELATE_EINVAL, "bad argument");
  if (count > 0 && (!cand_i || !cand_j || !cand_dist)) return fail(RELATE_EINVAL, "null candidate arrays");
  const double inf = std::numeric_limits<float>::infinity();  // Candidate's initial distances, src/tree_builder.hpp:15-17
  // bool operator>(const Candidate&, const Candidate&), src/tree_builder.cpp:8-23
  auto greater = [](const relate_candidate& a, const relate_candidate& b) {
    if (a.replace && a.dist3 >= b.dist3) {
      if (a.dist3 > b.dist3) return true;
      if (a.dist > b.dist || (a.dist == b.dist && a.dist2 > b.dist2)) return true;
    }
    return a.dist > b.dist || (a.dist == b.dist && a.dist2 > b.dist2);
  };
  for (int n = 0; n < N; n++) {  // :66-67 / :139-142
    mcandidates[n].dist = mcandidates[n].dist2 = inf;
    if (sample_ages) {
      mcandidates[n].dist3 = inf;
      mcandidates[n].replace = 0;
    }
  }
  for (int64_t c = 0; c < count; c++) {
    const int it = cand_i[c], jt = cand_j[c];
    if (it < 0 || jt <= it || jt >= N) return fail(RELATE_EINVAL, "candidate pair out of order or out of range");
    if (!sample_ages) {
      const float sym_dist = cand_dist[c];
      const float dist_random = (float)draw(draw_ctx);  // the reference keeps the draw in a float (src/tree_builder.hpp:47)
      for (relate_candidate* m : {&mcandidates[it], &mcandidates[jt]})  // :97-108
        if (m->dist > sym_dist || (m->dist == sym_dist && m->dist2 > dist_random)) {
          m->lin1 = it;
          m->lin2 = jt;
          m->dist = sym_dist;
          m->dist2 = dist_random;
        }
      const relate_candidate& mj = mcandidates[jt];
      if (best->dist > mj.dist || (best->dist == mj.dist && best->dist2 > mj.dist2)) {  // :109-114
        best->lin1 = it;
        best->lin2 = jt;
        best->dist = sym_dist;
        best->dist2 = mj.dist2;
      }
    } else {
      relate_candidate cand = {};  // :169-172
      cand.lin1 = cand.lin2 = -1;
      cand.dist = cand_dist[c];
      cand.dist3 = std::max(sample_ages[it], sample_ages[jt]);
      cand.dist2 = draw(draw_ctx);
      for (relate_candidate* m : {&mcandidates[it], &mcandidates[jt]})  // :173-192
        if ((m->dist == inf || cand.dist3 <= age) && greater(*m, cand)) {
          cand.replace = cand.dist3 > age ? 1 : 0;
          *m = cand;
          m->lin1 = it;
          m->lin2 = jt;
        }
      const relate_candidate& mj = mcandidates[jt];
      if ((best->dist == inf || mj.dist3 <= age) && greater(*best, mj)) {  // :193-200
        *best = mj;
        best->replace = best->dist3 > age ? 1 : 0;
      }
    }
  }
  return RELATE_OK;
}

int relate_plan_timings(relate_plan* p, float* ms_pack, float* ms_prepare, float* ms_paint)
{
  if (!p) return fail(RELATE_EINVAL, "null plan");
  HIP_TRY(hipSetDevice(p->device));
  float v[3] = {0.f, 0.f, 0.f};
  const bool have[3] = {p->have_pack, p->have_prepare, p->have_paint};
  for (int i = 0; i < 3; i++) {
    if (!have[i]) continue;
    HIP_TRY(hipEventSynchronize(p->ev[2 * i + 1]));
    HIP_TRY(hipEventElapsedTime(&v[i], p->ev[2 * i], p->ev[2 * i + 1]));
  }
  if (ms_pack) *ms_pack = v[0];
  if (ms_prepare) *ms_prepare = v[1];
  if (ms_paint) *ms_paint = v[2];
  return RELATE_OK;
}

int relate_plan_geometry(const relate_plan* p, int* waves, int* mpl)
{
  if (!p) return fail(RELATE_EINVAL, "null plan");
  if (waves) *waves = p->geo.G;
  if (mpl) *mpl = p->geo.MPL;
  return RELATE_OK;
}

int relate_plan_repaint_geometry(const relate_plan* p, int* waves, int* mpl, int* checkpoint_rows)
{
  if (!p) return fail(RELATE_EINVAL, "null plan");
  if (!p->have_prepare || p->geo_repaint.MPL <= 0) return fail(RELATE_EINVAL, "no targets prepared");
  if (waves) *waves = p->geo_repaint.G;
  if (mpl) *mpl = p->geo_repaint.MPL;
  if (checkpoint_rows) *checkpoint_rows = relate::repaint_block_rows(p->geo_repaint.MPL);
  return RELATE_OK;
}

int relate_fast_log(const float* in, float* out, int n, int device)
{
  if (!in || !out || n < 0) return fail(RELATE_EINVAL, "bad argument");
  if (relate_device_count() <= 0) return fail(RELATE_EGPU, "no HIP device visible");
  if (n == 0) return RELATE_OK;
  HIP_TRY(hipSetDevice(device));
  DevBuf a, b;
  if (int rc = a.reserve(sizeof(float) * (size_t)n)) return rc;
  if (int rc = b.reserve(sizeof(float) * (size_t)n)) {
    a.release();
    return rc;
  }
  hipError_t e = hipMemcpy(a.p, in, sizeof(float) * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = relate::launch_fast_log(a.as<float>(), b.as<float>(), n, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out, b.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost);
  a.release();
  b.release();
  if (e != hipSuccess) return fail(RELATE_EGPU, hipGetErrorString(e));
  return RELATE_OK;
}

}  // extern "C"

// --------------------------------------------------------------------------
// Batched painting of a target range on one device, records handed to a sink.
// --------------------------------------------------------------------------
namespace {

// Pinned staging memory for the copy back of the records.  Pinning a few hundred megabytes costs as much as
// painting a chunk, so one block (at most kStagingBytes) is kept and handed to one caller at a time; a second
// concurrent caller pins a block of its own and frees it when done.  relate_release_host_staging() frees the
// kept block.
constexpr size_t kStagingBytes = (size_t)1 << 30;
struct PinnedCache {
  std::mutex m;
  void* p = nullptr;
  size_t bytes = 0;
  bool busy = false;
};
PinnedCache g_pinned;

struct PinnedLease {
  void* p = nullptr;
  bool cached = false;
  int acquire(size_t bytes)
  {
    {
      std::lock_guard<std::mutex> lk(g_pinned.m);
      if (!g_pinned.busy) {
        if (g_pinned.bytes < bytes) {
          if (g_pinned.p) (void)hipHostFree(g_pinned.p);
          g_pinned.p = nullptr;
          g_pinned.bytes = 0;
          if (hipHostMalloc(&g_pinned.p, bytes, hipHostMallocDefault) != hipSuccess) {
            g_pinned.p = nullptr;
            return fail(RELATE_ENOMEM, "pinned host allocation of " + std::to_string(bytes) + " bytes failed");
          }
          g_pinned.bytes = bytes;
        }
        g_pinned.busy = true;
        p = g_pinned.p;
        cached = true;
        return RELATE_OK;
      }
    }
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
      p = nullptr;
      return fail(RELATE_ENOMEM, "pinned host allocation of " + std::to_string(bytes) + " bytes failed");
    }
    return RELATE_OK;
  }
  ~PinnedLease()
  {
    if (cached) {
      std::lock_guard<std::mutex> lk(g_pinned.m);
      g_pinned.busy = false;
    } else if (p) {
      (void)hipHostFree(p);
    }
  }
};

// nd_k of targets [k_begin, k_end) of the loaded chunk, on the host.
int derived_counts(relate_plan* p, int k_begin, int k_end, std::vector<int32_t>& nd)
{
  const int K = k_end - k_begin;
  nd.resize((size_t)K);
  DevBuf d;
  if (int rc = d.reserve(sizeof(int32_t) * (size_t)K)) return rc;
  hipError_t e = relate::launch_count_derived(p->d_cols, p->col_stride, p->L, k_begin, K, d.as<int32_t>(), p->own_stream);
  if (e == hipSuccess) e = hipMemcpyAsync(nd.data(), d.p, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, p->own_stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->own_stream);
  d.release();
  if (e != hipSuccess) return fail(RELATE_EGPU, std::string("count_derived: ") + hipGetErrorString(e));
  return RELATE_OK;
}

// Contiguous partition of targets into `parts` ranges whose sums of nd (the number of sweeps a target costs)
// are as equal as a contiguous split allows; every range is non-empty while there are targets left.
void shard_by_work(const int32_t* nd, int N, int parts, int32_t* bounds)
{
  std::vector<int64_t> prefix((size_t)N + 1, 0);
  for (int k = 0; k < N; k++) prefix[k + 1] = prefix[k] + std::max<int32_t>(nd[k], 0);
  const int64_t total = prefix[N];
  bounds[0] = 0;
  for (int i = 1; i < parts; i++) {
    const int64_t want = (int64_t)((__int128)total * i / parts);
    int k = (int)(std::lower_bound(prefix.begin(), prefix.end(), want) - prefix.begin());
    if (k > 0 && want - prefix[k - 1] <= prefix[k] - want) k--;  // the nearer of the two cuts
    const int lo = std::min(N, bounds[i - 1] + 1), hi = std::max(lo, N - (parts - i));
    bounds[i] = std::min(std::max(k, lo), hi);
  }
  bounds[parts] = N;
}

// Batches of consecutive targets: the records of a batch (W * R bytes per target) and its site lists (28 bytes
// per derived site) stay under `budget` bytes of HBM, and one window's slab of records under `slab_cap` bytes.
std::vector<int> plan_batches(const std::vector<int32_t>& nd, int k_begin, size_t R, int W, size_t budget, size_t slab_cap)
{
  std::vector<int> cuts{k_begin};
  size_t bytes = 0, slab = 0;
  for (size_t i = 0; i < nd.size(); i++) {
    const size_t t = R * (size_t)W + (size_t)28 * (size_t)nd[i];
    if (i > (size_t)(cuts.back() - k_begin) && (bytes + t > budget || slab + R > slab_cap)) {
      cuts.push_back(k_begin + (int)i);
      bytes = 0;
      slab = 0;
    }
    bytes += t;
    slab += R;
  }
  cuts.push_back(k_begin + (int)nd.size());
  return cuts;
}

// Host threads that carry slabs from the pinned ring to their destination.  A thread that copies into fresh
// pageable memory moves about 8 GB/s (it takes the page faults), the link five to six times that: half of
// the cores this process may run on, between 4 and 12.  RELATE_COPY_THREADS overrides it (measurements).
int copy_back_threads()
{
  static const int n = [] {
    if (const char* e = std::getenv("RELATE_COPY_THREADS")) {
      const int v = std::atoi(e);
      if (v >= 1 && v <= 64) return v;
    }
    cpu_set_t set;
    int cores = 8;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cores = CPU_COUNT(&set);
    return std::max(4, std::min(12, cores / 2));
  }();
  return n;
}

// sink(w, k0, nk, ptr): the records of targets [k0, k0+nk) of window w, contiguous at ptr (host memory).
// The records come back window by window through a ring of pinned slots, and a few host threads hand each
// window's slab to the sink as soon as it has arrived: the sink (a memcpy into pageable memory, or pwrite) is
// slower than the link, and the slabs of different windows go to different places.
// With several batches of targets the copy-back of a batch runs beside the painting of the next one: two record
// buffers in HBM, the copies on a stream of their own behind an event of the painting stream, and a host thread
// per batch that feeds the ring while the calling thread prepares and paints the next batch.
template <class Sink>
int paint_range(relate_plan* p, double theta_p, double theta_d, int k_begin, int k_end, Sink&& sink)
{
  const int N = p->N, W = p->W;
  const size_t R = relate_record_bytes(N);
  HIP_TRY(hipSetDevice(p->device));
  std::vector<int32_t> nd;
  if (int rc = derived_counts(p, k_begin, k_end, nd)) return rc;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t slab_cap = std::max(kStagingBytes / 2, R);  // two slots at least (one target per slot at worst)
  int rc = RELATE_OK;
  DevBuf d_out[2];
  std::vector<int> cuts;
  int widest = 0, n_buf = 1;
  // batches of at most 12 GiB of records and lists (a quarter of the free memory at most), two of them resident: the
  // link carries a batch in a fifth of a second, during which the next one is painted, so that only the painting of
  // the first batch is not hidden.  A refused allocation halves the budget and plans the batches again.
  size_t budget = std::min<size_t>(free_b / 4, (size_t)12 << 30);
  if (const char* e = std::getenv("RELATE_BATCH_BYTES")) {  // tests: several batches of a small chunk
    const long long v = std::atoll(e);
    if (v > 0) budget = std::min<size_t>(budget, (size_t)v);
  }
  for (int attempt = 0;; attempt++) {
    cuts = plan_batches(nd, k_begin, R, W, budget, slab_cap);
    n_buf = cuts.size() > 2 ? 2 : 1;
    widest = 0;
    for (size_t b = 0; b + 1 < cuts.size(); b++) widest = std::max(widest, cuts[b + 1] - cuts[b]);
    rc = RELATE_OK;
    for (int i = 0; i < n_buf && rc == RELATE_OK; i++) rc = d_out[i].reserve(R * (size_t)W * (size_t)widest);
    if (rc == RELATE_OK || attempt == 6 || widest == 1) break;
    d_out[0].release();
    d_out[1].release();
    budget /= 2;
  }
  if (rc) return rc;
  const size_t slot_bytes = R * (size_t)widest;
  const int n_slots = (int)std::max<size_t>(2, std::min<size_t>((size_t)W, std::max<size_t>(kStagingBytes, 2 * slot_bytes) / slot_bytes));
  PinnedLease lease;
  hipStream_t copy_stream = nullptr;
  hipEvent_t painted[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> arrived(n_slots, nullptr);
  auto cleanup = [&]() {
    for (auto& e : arrived)
      if (e) (void)hipEventDestroy(e);
    for (auto& e : painted)
      if (e) (void)hipEventDestroy(e);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    d_out[0].release();
    d_out[1].release();
  };
  if (int rc2 = lease.acquire(slot_bytes * (size_t)n_slots)) {
    cleanup();
    return rc2;
  }
  uint8_t* h_ring = static_cast<uint8_t*>(lease.p);
  {
    hipError_t e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&painted[i], hipEventDisableTiming);
    for (int i = 0; i < n_slots && e == hipSuccess; i++) e = hipEventCreateWithFlags(&arrived[i], hipEventDisableTiming);
    if (e != hipSuccess) {
      cleanup();
      return fail(RELATE_EGPU, std::string("stream / event creation: ") + hipGetErrorString(e));
    }
  }
  const int n_threads = std::min(copy_back_threads(), std::min(W, n_slots));

  // The copy-back of one painted batch: the windows through the ring on copy_stream, drained by n_threads host
  // threads.  Runs on a thread of its own; its result and message are left in `crc` / `cerr`.
  int crc = RELATE_OK;
  std::string cerr;
  auto copy_back = [&](int k0, int nk, const uint8_t* d_records, hipEvent_t ready) {
    if (hipSetDevice(p->device) != hipSuccess || hipStreamWaitEvent(copy_stream, ready, 0) != hipSuccess) {
      crc = RELATE_EGPU;
      cerr = "copy-back could not be ordered behind the painting";
      return;
    }
    const size_t slab = R * (size_t)nk;
    // window w travels through slot w % n_slots: issued[w] once its copy is on the stream, drained[w] once the
    // sink has returned (the slot may then take window w + n_slots)
    std::mutex m;
    std::condition_variable cv;
    std::vector<char> issued((size_t)W, 0), drained((size_t)W, 0);
    bool failed = false;
    std::vector<int> rcs(n_threads, RELATE_OK);
    std::vector<std::string> errs(n_threads);
    auto drain = [&](int ti) {
      if (hipSetDevice(p->device) != hipSuccess) {
        rcs[ti] = RELATE_EGPU;
        errs[ti] = "hipSetDevice failed in a copy-back thread";
      }
      for (int w = ti; w < W && rcs[ti] == RELATE_OK; w += n_threads) {
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return issued[w] || failed; });
          if (failed) return;
        }
        const int slot = w % n_slots;
        const hipError_t ew = hipEventSynchronize(arrived[slot]);
        if (ew != hipSuccess) {
          rcs[ti] = RELATE_EGPU;
          errs[ti] = std::string("painting or copy-back failed: ") + hipGetErrorString(ew);
        } else {
          const int r = sink(w, k0, nk, h_ring + (size_t)slot * slot_bytes);
          if (r != RELATE_OK) {
            rcs[ti] = r;
            errs[ti] = g_err;  // the sink's message lives in this thread
          }
        }
        std::lock_guard<std::mutex> lk(m);
        drained[w] = 1;
        if (rcs[ti] != RELATE_OK) failed = true;
        cv.notify_all();
      }
      if (rcs[ti] != RELATE_OK) {
        std::lock_guard<std::mutex> lk(m);
        failed = true;
        cv.notify_all();
      }
    };
    std::vector<std::thread> th;
    for (int ti = 0; ti < n_threads; ti++) th.emplace_back(drain, ti);
    hipError_t e = hipSuccess;
    for (int w = 0; w < W && e == hipSuccess; w++) {
      const int slot = w % n_slots;
      if (w >= n_slots) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return drained[w - n_slots] || failed; });
        if (failed) break;
      }
      e = hipMemcpyAsync(h_ring + (size_t)slot * slot_bytes, d_records + (size_t)w * slab, slab, hipMemcpyDeviceToHost, copy_stream);
      if (e == hipSuccess) e = hipEventRecord(arrived[slot], copy_stream);
      std::lock_guard<std::mutex> lk(m);
      if (e == hipSuccess) issued[w] = 1;
      else failed = true;
      cv.notify_all();
    }
    for (auto& t : th) t.join();
    // the ring and the record buffer are reused: everything must have left the stream
    const hipError_t es = hipStreamSynchronize(copy_stream);
    if (e != hipSuccess) {
      crc = RELATE_EGPU;
      cerr = std::string("copy-back failed: ") + hipGetErrorString(e);
    }
    for (int ti = 0; ti < n_threads; ti++)
      if (rcs[ti] != RELATE_OK && crc == RELATE_OK) {
        crc = rcs[ti];
        cerr = errs[ti];
      }
    if (es != hipSuccess && crc == RELATE_OK) {
      crc = RELATE_EGPU;
      cerr = std::string("painting or copy-back failed: ") + hipGetErrorString(es);
    }
  };

  std::thread copier;
  auto join_copier = [&]() {
    if (copier.joinable()) copier.join();
    if (crc != RELATE_OK && rc == RELATE_OK) rc = fail(crc, cerr);
  };
  for (size_t b = 0; b + 1 < cuts.size() && rc == RELATE_OK; b++) {
    const int k0 = cuts[b], k1 = cuts[b + 1], nk = k1 - k0;
    const int buf = n_buf == 2 ? (int)(b & 1) : 0;
    // the copy-back of batch b-1 (the other buffer) may still run; that of batch b-2 (this buffer) has been joined
    rc = relate_plan_prepare(p, theta_p, theta_d, k0, k1, nullptr, p->own_stream);
    if (rc == RELATE_ENOMEM && nk > 1) {
      // the site lists did not fit beside the records: split this batch (the buffers are wide enough for a part)
      cuts.insert(cuts.begin() + (long)b + 1, k0 + nk / 2);
      if (n_buf == 1) {  // a second batch after all: it reuses the one buffer, after the copy-back of the first
        join_copier();
      }
      rc = RELATE_OK;
      b--;
      continue;
    }
    if (rc) break;
    if (n_buf == 1) join_copier();  // one buffer: the previous batch must have left it
    if (rc) break;
    rc = relate_plan_paint(p, d_out[buf].p, p->own_stream);
    if (rc) break;
    if (hipEventRecord(painted[buf], p->own_stream) != hipSuccess) {
      rc = fail(RELATE_EGPU, "hipEventRecord failed");
      break;
    }
    join_copier();  // the ring serves one batch at a time
    if (rc) break;
    copier = std::thread(copy_back, k0, nk, static_cast<const uint8_t*>(d_out[buf].p), painted[buf]);
  }
  join_copier();
  (void)hipStreamSynchronize(p->own_stream);
  cleanup();
  return rc;
}

// pwrite of the whole buffer at `offset` (short writes continued)
int pwrite_all(int fd, const uint8_t* src, size_t total, size_t offset, int w)
{
  size_t done = 0;
  while (done < total) {
    const ssize_t n = pwrite(fd, src + done, total - done, (off_t)(offset + done));
    if (n <= 0) return fail(RELATE_EIO, "write to relate_" + std::to_string(w) + ".bin failed: " + std::strerror(errno));
    done += (size_t)n;
  }
  return RELATE_OK;
}

}  // namespace

extern "C" void relate_release_host_staging(void)
{
  std::lock_guard<std::mutex> lk(g_pinned.m);
  if (g_pinned.busy || !g_pinned.p) return;
  (void)hipHostFree(g_pinned.p);
  g_pinned.p = nullptr;
  g_pinned.bytes = 0;
}

extern "C" int relate_plan_derived_counts(relate_plan* p, int32_t* nd)
{
  if (!p || !nd) return fail(RELATE_EINVAL, "null argument");
  if (!p->d_rows) return fail(RELATE_EINVAL, "no chunk loaded");
  HIP_TRY(hipSetDevice(p->device));
  std::vector<int32_t> v;
  if (int rc = derived_counts(p, 0, p->N, v)) return rc;
  std::memcpy(nd, v.data(), sizeof(int32_t) * v.size());
  return RELATE_OK;
}

extern "C" int relate_shard_by_work(int N, const int32_t* nd, int parts, int32_t* bounds)
{
  if (N < 1 || parts < 1 || !nd || !bounds) return fail(RELATE_EINVAL, "bad argument");
  shard_by_work(nd, N, parts, bounds);
  return RELATE_OK;
}

extern "C" int relate_plan_paint_to_files(relate_plan* p, double theta_p, double theta_d, int k_begin, int k_end,
                                          const int* fds, int num_fds)
{
  if (!p || !fds) return fail(RELATE_EINVAL, "null argument");
  if (!p->d_rows) return fail(RELATE_EINVAL, "no chunk loaded");
  if (num_fds != p->W) return fail(RELATE_EINVAL, "one file descriptor per window is needed");
  if (k_begin < 0 || k_end > p->N || k_begin > k_end) return fail(RELATE_EINVAL, "bad target range");
  if (k_begin == k_end) return RELATE_OK;
  const size_t R = relate_record_bytes(p->N);
  return paint_range(p, theta_p, theta_d, k_begin, k_end, [&](int w, int k0, int nk, const uint8_t* src) {
    return pwrite_all(fds[w], src, (size_t)nk * R, (size_t)k0 * R, w);
  });
}

extern "C" int relate_paint_stones(int N, int L, const uint8_t* seq, const double* r, double theta_p, double theta_d,
                                   int W, const int32_t* wb, int k_begin, int k_end, uint8_t* out, int device)
{
  if (!seq || !r || !wb || !out) return fail(RELATE_EINVAL, "null argument");
  if (k_begin < 0 || k_end > N || k_begin > k_end) return fail(RELATE_EINVAL, "bad target range");
  if (k_begin == k_end) return RELATE_OK;
  relate_plan* p = nullptr;
  if (int rc = relate_plan_create(device < 0 ? 0 : device, &p)) return rc;
  int rc = relate_plan_load_host(p, N, L, seq, r, W, wb);
  const size_t R = relate_record_bytes(N);
  const size_t K = (size_t)(k_end - k_begin);
  if (!rc)
    rc = paint_range(p, theta_p, theta_d, k_begin, k_end, [&](int w, int k0, int nk, const uint8_t* src) {
      std::memcpy(out + ((size_t)w * K + (size_t)(k0 - k_begin)) * R, src, (size_t)nk * R);
      return RELATE_OK;
    });
  relate_plan_destroy(p);
  return rc;
}

// --------------------------------------------------------------------------
// Data / painter mirror
// --------------------------------------------------------------------------
// One painted batch of targets, kept with the data it was computed from: the Rust caller builds a
// new painter for every target (src/pipelines.rs:179-194), so the painter cannot own it.
struct StoneCache {
  double theta_p = 0, theta_d = 0;
  std::vector<int32_t> wb;
  int k0 = 0, k1 = 0;
  std::vector<uint8_t> records;  // [W][k1-k0][R]
};

struct relate_data {
  int N = 0, L = 0, Ne = 30000;
  double mu = 1.25e-8, theta = 0.001, ntheta = 0.999;
  std::vector<uint8_t> seq;
  std::vector<double> r;
  mutable StoneCache cache;
  mutable std::mutex cache_mutex;
};

struct relate_painter {
  size_t N = 0;
  double theta = 0.001;
  int device = 0;
};

extern "C" {

int relate_data_new(const char* hap, const char* pos, const char* dist, const char* rec, const char* rpos,
                    const char* state, int Ne, double mu, relate_data** out)
{
  (void)pos;
  (void)dist;
  (void)rpos;
  (void)state;
  if (!hap || !rec || !out) return fail(RELATE_EINVAL, "the .hap and .r file names are required");
  *out = nullptr;
  relate_data* d = new relate_data();
  d->Ne = Ne;
  d->mu = mu;
  std::string err;
  if (!relate::io::read_hap(hap, d->N, d->L, d->seq, err) || !relate::io::read_vector(rec, d->r, err)) {
    delete d;
    return fail(RELATE_EIO, err);
  }
  if ((int)d->r.size() != d->L) {
    delete d;
    return fail(RELATE_EINVAL, "the .r file does not have one entry per SNP of the .hap file");
  }
  *out = d;
  return RELATE_OK;
}

int relate_data_set_painting(relate_data* d, double theta, double rho)
{
  if (!d) return fail(RELATE_EINVAL, "null data");
  if (!(theta > 0.0 && theta < 1.0)) return fail(RELATE_EINVAL, "theta must lie in (0,1)");
  d->theta = theta;
  d->ntheta = 1.0 - theta;
  for (double& v : d->r) v *= rho;
  d->cache = StoneCache();  // stones painted from the old r and theta are stale
  return RELATE_OK;
}

int relate_data_dims(const relate_data* d, int* N, int* L)
{
  if (!d) return fail(RELATE_EINVAL, "null data");
  if (N) *N = d->N;
  if (L) *L = d->L;
  return RELATE_OK;
}

void relate_data_free(relate_data* d) { delete d; }

int relate_painter_new(size_t N, double theta, int device, relate_painter** out)
{
  if (!out) return fail(RELATE_EINVAL, "null output pointer");
  *out = nullptr;
  if (N < 2) return fail(RELATE_EINVAL, "need at least two haplotypes");
  if (!(theta > 0.0 && theta < 1.0)) return fail(RELATE_EINVAL, "theta must lie in (0,1) (src/fast_painting.hpp:23)");
  relate_painter* p = new relate_painter();
  p->N = N;
  p->theta = theta;
  p->device = device < 0 ? 0 : device;
  *out = p;
  return RELATE_OK;
}

void relate_painter_free(relate_painter* p) { delete p; }

int relate_painter_paint_stepping_stones(relate_painter* p, const relate_data* d, const char* basename, size_t W,
                                         const int32_t* wb, int k)
{
  if (!p || !d || !basename || !wb) return fail(RELATE_EINVAL, "null argument");
  if ((size_t)d->N != p->N) return fail(RELATE_EINVAL, "painter and data disagree on N");
  if (k < 0 || k >= d->N) return fail(RELATE_EINVAL, "target index out of range");
  const size_t R = relate_record_bytes(d->N);
  std::lock_guard<std::mutex> guard(d->cache_mutex);
  StoneCache& c = d->cache;
  const bool hit = c.theta_p == p->theta && c.theta_d == d->theta && c.wb.size() == W + 1 &&
                   std::equal(c.wb.begin(), c.wb.end(), wb) && k >= c.k0 && k < c.k1;
  if (!hit) {
    // paint a batch starting at k: the caller walks k upwards (src/pipelines.rs:179-194)
    const size_t budget = (size_t)2 << 30;
    const int nk = (int)std::min<size_t>((size_t)(d->N - k), std::max<size_t>(1, budget / (R * std::max<size_t>(W, 1))));
    std::vector<uint8_t> rec((size_t)W * nk * R);
    int rc = relate_paint_stones(d->N, d->L, d->seq.data(), d->r.data(), p->theta, d->theta, (int)W, wb, k, k + nk,
                                 rec.data(), p->device);
    if (rc) return rc;
    c.records.swap(rec);
    c.theta_p = p->theta;
    c.theta_d = d->theta;
    c.wb.assign(wb, wb + W + 1);
    c.k0 = k;
    c.k1 = k + nk;
  }
  const size_t nk = (size_t)(c.k1 - c.k0);
  for (size_t w = 0; w < W; w++) {
    char name[1024];
    snprintf(name, sizeof name, "%s_%i.bin", basename, (int)w);
    FILE* f = fopen(name, "ab");  // append: the order of the calls is the file format (src/fast_painting.cpp:645)
    if (!f) return fail(RELATE_EIO, std::string("cannot open ") + name);
    const uint8_t* src = c.records.data() + (w * nk + (size_t)(k - c.k0)) * R;
    const bool ok = fwrite(src, 1, R, f) == R;
    if (fclose(f) != 0 || !ok) return fail(RELATE_EIO, std::string("short write to ") + name);
  }
  return RELATE_OK;
}

// --------------------------------------------------------------------------
// Whole stage
// --------------------------------------------------------------------------
}  // extern "C"

namespace {

// The window files of one chunk while they are being written.  Consumers trust the size of relate_<w>.bin
// (N records, src/anc_builder.cpp:65-69) and later stages delete their inputs, so a file must be complete or
// absent: records go to relate_<w>.bin.part, which commit() renames once every record of every window has
// been written; whatever is left uncommitted is unlinked.  Stale relate_<w>.bin of an earlier run are removed
// at the start, as the reference's fopen(..., "wb") (pipeline/Paint.cpp:74) would have emptied them.
struct WindowFiles {
  std::vector<std::string> finals, parts;
  std::vector<int> fds;
  bool committed = false;
  int open_all(const std::string& paint_dir, int W)
  {
    for (int w = 0; w < W; w++) {
      finals.push_back(paint_dir + "/relate_" + std::to_string(w) + ".bin");
      parts.push_back(finals.back() + ".part");
      if (unlink(finals.back().c_str()) != 0 && errno != ENOENT) return fail(RELATE_EIO, "cannot remove stale " + finals.back());
      const int fd = open(parts.back().c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
      if (fd < 0) return fail(RELATE_EIO, "cannot open " + parts.back() + ": " + std::strerror(errno));
      fds.push_back(fd);
    }
    return RELATE_OK;
  }
  int commit()
  {
    int rc = RELATE_OK;
    for (size_t w = 0; w < fds.size(); w++)
      if (fds[w] >= 0) {
        if (close(fds[w]) != 0 && rc == RELATE_OK) rc = fail(RELATE_EIO, "close of " + parts[w] + " failed");
        fds[w] = -1;
      }
    for (size_t w = 0; w < parts.size() && rc == RELATE_OK; w++)
      if (rename(parts[w].c_str(), finals[w].c_str()) != 0) rc = fail(RELATE_EIO, "cannot rename " + parts[w]);
    if (rc != RELATE_OK) {
      const std::string saved = g_err;
      for (const std::string& f : finals) (void)unlink(f.c_str());  // some were renamed already: all or nothing
      g_err = saved;
      return rc;
    }
    committed = true;
    return RELATE_OK;
  }
  ~WindowFiles()
  {
    const std::string saved = g_err;
    for (int fd : fds)
      if (fd >= 0) (void)close(fd);
    if (!committed)
      for (const std::string& f : parts) (void)unlink(f.c_str());
    g_err = saved;
  }
};

// Makes `p` (on another device) paint the chunk that `src` holds: the packed matrix and r travel from device
// to device (8x smaller than the byte matrix, packed once), nothing is packed again.
int plan_adopt_copy(relate_plan* p, const relate_plan* src)
{
  HIP_TRY(hipSetDevice(p->device));
  const size_t bytes = rows_bytes(src) + cols_bytes(src);
  if (p->packed_borrowed) {
    p->packed = DevBuf();
    p->packed_borrowed = false;
  }
  if (int rc = p->packed.reserve(bytes)) return rc;
  if (int rc = p->r_own.reserve(sizeof(double) * (size_t)src->L)) return rc;
  HIP_TRY(hipMemcpyPeerAsync(p->packed.p, p->device, src->d_rows, src->device, bytes, p->own_stream));
  HIP_TRY(hipMemcpyPeerAsync(p->r_own.p, p->device, src->d_r, src->device, sizeof(double) * (size_t)src->L, p->own_stream));
  if (int rc = plan_set_chunk(p, src->N, src->L, src->W, src->wb.data(), p->own_stream)) return rc;
  HIP_TRY(hipStreamSynchronize(p->own_stream));
  p->d_rows = p->packed.as<uint64_t>();
  p->d_cols = reinterpret_cast<const uint64_t*>(static_cast<const char*>(p->packed.p) + rows_bytes(src));
  p->d_r = p->r_own.as<double>();
  p->have_pack = false;
  return RELATE_OK;
}

}  // namespace

extern "C" {

int relate_paint_chunk(const char* output_dir, int chunk_index, const double* theta, const double* rho, int device)
{
  if (!output_dir) return fail(RELATE_EINVAL, "null output directory");
  const int ndev_all = relate_device_count();
  if (ndev_all <= 0) return fail(RELATE_EGPU, "no HIP device visible: the painting path has no CPU fallback");
  if (device >= ndev_all) return fail(RELATE_EINVAL, "device index out of range");
  const std::string dir(output_dir), c = std::to_string(chunk_index);
  std::string err;
  int N = 0, L = 0;
  std::vector<int32_t> wb;
  if (!relate::io::read_parameters(dir + "/parameters_c" + c + ".bin", N, L, wb, err)) return fail(RELATE_EIO, err);
  const int W = (int)wb.size() - 1;
  relate_data* d = nullptr;
  if (int rc = relate_data_new((dir + "/chunk_" + c + ".hap").c_str(), nullptr, nullptr,
                               (dir + "/chunk_" + c + ".r").c_str(), nullptr, nullptr, 30000, 1.25e-8, &d))
    return rc;
  std::unique_ptr<relate_data, void (*)(relate_data*)> d_guard(d, relate_data_free);
  if (d->N != N || d->L != L) return fail(RELATE_EINVAL, "parameters_c<c>.bin and chunk_<c>.hap disagree on N or L");
  if (theta || rho)
    if (int rc = relate_data_set_painting(d, theta ? *theta : d->theta, rho ? *rho : 1.0)) return rc;
  fprintf(stderr, "---------------------------------------------------------\n");
  fprintf(stderr, "Painting sequences...\n");  // pipeline/Paint.cpp:60-61

  const std::string chunk_dir = dir + "/chunk_" + c, paint_dir = chunk_dir + "/paint";
  if ((mkdir(chunk_dir.c_str(), 0777) != 0 && errno != EEXIST) || (mkdir(paint_dir.c_str(), 0777) != 0 && errno != EEXIST))
    return fail(RELATE_EIO, "cannot create " + paint_dir);
  WindowFiles files;
  if (int rc = files.open_all(paint_dir, W)) return rc;

  // contiguous target ranges, one per device; every record has a fixed place k*R in its file
  // device < 0: all visible devices, or the list in RELATE_DEVICES ("0,2,3"; an index may repeat, "0,0" runs two
  // plans, two host threads and the device-to-device hand-over on one GPU)
  std::vector<int> devs;
  if (device >= 0) {
    devs.push_back(device);
  } else if (const char* list = getenv("RELATE_DEVICES"); list && *list) {
    for (const char* q = list; *q;) {
      char* end = nullptr;
      const long v = strtol(q, &end, 10);
      if (end == q || v < 0 || v >= ndev_all || (*end && *end != ','))
        return fail(RELATE_EINVAL, std::string("RELATE_DEVICES: expected device indices below ") + std::to_string(ndev_all) +
                                       " separated by commas, got \"" + list + "\"");
      devs.push_back((int)v);
      q = *end ? end + 1 : end;
    }
    if (devs.empty() || devs.size() > 64) return fail(RELATE_EINVAL, "RELATE_DEVICES names no device or more than 64");
  } else {
    for (int i = 0; i < ndev_all; i++) devs.push_back(i);
  }
  const int nd = (int)std::min<size_t>(devs.size(), (size_t)N);
  // the first device packs; the others receive the packed matrix from it
  relate_plan* first = nullptr;
  if (int rc = relate_plan_create(devs[0], &first)) return rc;
  std::unique_ptr<relate_plan, void (*)(relate_plan*)> first_guard(first, relate_plan_destroy);
  if (int rc = relate_plan_load_host(first, N, L, d->seq.data(), d->r.data(), W, wb.data())) return rc;
  std::vector<int32_t> bounds((size_t)nd + 1, 0);
  bounds[nd] = N;
  if (nd > 1) {
    // ranges of equal work: the cost of a target is its number of derived sites
    std::vector<int32_t> counts;
    if (int rc = derived_counts(first, 0, N, counts)) return rc;
    shard_by_work(counts.data(), N, nd, bounds.data());
  }
  std::vector<int> rcs(nd, RELATE_OK);
  std::vector<std::string> errs(nd);
  const double th = d->theta;
  auto work = [&](int i) {
    relate_plan* p = first;
    int r = RELATE_OK;
    if (i > 0) {
      p = nullptr;
      r = relate_plan_create(devs[i], &p);
      if (!r) r = plan_adopt_copy(p, first);
    }
    if (!r) r = relate_plan_paint_to_files(p, th, th, bounds[i], bounds[i + 1], files.fds.data(), W);
    if (i > 0) relate_plan_destroy(p);
    rcs[i] = r;
    errs[i] = g_err;
  };
  if (nd == 1) {
    work(0);
  } else {
    std::vector<std::thread> thr;
    for (int i = 0; i < nd; i++) thr.emplace_back(work, i);
    for (auto& t : thr) t.join();
  }
  for (int i = 0; i < nd; i++)
    if (rcs[i] != RELATE_OK) return fail(rcs[i], errs[i]);  // `files` unlinks the partial files
  return files.commit();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// DistanceMeasure (src/anc_builder.hpp:35-90, src/anc_builder.cpp:9-211) with the topology of the current
// section and the matrix in HBM: the object AncesTreeBuilder::BuildTopology asks for the distance matrix of
// a SNP.  What stays on the host is the reference's own bookkeeping, under its own names.
struct relate_distance {
  relate_plan* plan = nullptr;
  const uint8_t* d_stones = nullptr;
  int N = 0, L = 0;
  int words = 0;                  // 64-bit words per SNP of `derived`
  std::vector<uint64_t> derived;  // bit n of row s: haplotype n carries the derived allele at SNP s
  std::vector<double> rpos;       // Data::rpos, L + 1
  int section = 0, section_startpos = -1, section_endpos = -1;
  std::vector<int64_t> v_snp_prev;
  std::vector<double> v_rpos_prev, v_rpos_next;
  DevBuf topology, logscales;
  // host copy of the section's logscales (the exponentials of GetMatrix, :171-172, are taken on the host with
  // the C library the reference uses); fetched once per re-painting, `ls_fetched` is recorded behind the copy
  HostBuf h_logscales;
  hipEvent_t ls_fetched = nullptr;
  bool ls_pending = false;
  int64_t rows = 0;
  int pitch = 0;
  std::vector<int64_t> row_offsets, row_index;
  std::vector<uint8_t> interpolate;
  std::vector<double> weight_left, weight_right;
  bool is_derived(int s, int n) const { return (derived[(size_t)s * words + (n >> 6)] >> (n & 63)) & 1ull; }
};

extern "C" {

int relate_distance_new(relate_plan* plan, const uint8_t* seq, const double* rpos, const void* d_stones, int first_section,
                        relate_distance** out)
{
  if (!plan || !seq || !rpos || !d_stones || !out) return fail(RELATE_EINVAL, "null argument");
  *out = nullptr;
  if (!plan->d_rows) return fail(RELATE_EINVAL, "no chunk loaded");
  if (plan->K != plan->N || plan->k_begin != 0) return fail(RELATE_EINVAL, "the plan must be prepared for all N targets");
  if (first_section < 0 || first_section >= plan->W) return fail(RELATE_EINVAL, "no such section");
  std::unique_ptr<relate_distance> d(new relate_distance());
  d->plan = plan;
  d->d_stones = static_cast<const uint8_t*>(d_stones);
  d->N = plan->N;
  d->L = plan->L;
  d->words = (d->N + 63) / 64;
  d->section = first_section;
  try {
    d->derived.assign((size_t)d->L * d->words, 0);
    for (int s = 0; s < d->L; s++) {
      const uint8_t* row = seq + (size_t)s * d->N;
      uint64_t* o = d->derived.data() + (size_t)s * d->words;
      for (int n = 0; n < d->N; n++)
        if (row[n] == '1') o[n >> 6] |= 1ull << (n & 63);
    }
    d->rpos.assign(rpos, rpos + d->L + 1);
    d->v_snp_prev.assign((size_t)d->N, 0);
    d->v_rpos_prev.assign((size_t)d->N, 0.0);
    d->v_rpos_next.assign((size_t)d->N, 0.0);
    d->row_offsets.assign((size_t)d->N + 1, 0);
    d->row_index.assign((size_t)d->N, 0);
    d->interpolate.assign((size_t)d->N, 0);
    d->weight_left.assign((size_t)d->N, 0.0);
    d->weight_right.assign((size_t)d->N, 0.0);
  } catch (const std::bad_alloc&) {
    return fail(RELATE_ENOMEM, "out of host memory");
  }
  HIP_TRY(hipSetDevice(plan->device));
  HIP_TRY(hipEventCreateWithFlags(&d->ls_fetched, hipEventDisableTiming));
  *out = d.release();
  return RELATE_OK;
}

void relate_distance_free(relate_distance* d)
{
  if (!d) return;
  (void)hipSetDevice(d->plan->device);
  if (d->ls_pending) (void)hipEventSynchronize(d->ls_fetched);
  d->topology.release();
  d->logscales.release();
  d->h_logscales.release();
  if (d->ls_fetched) (void)hipEventDestroy(d->ls_fetched);
  delete d;
}

int relate_distance_get_topology_with_repaint(relate_distance* d, int snp, void* stream)
{
  if (!d) return fail(RELATE_EINVAL, "null argument");
  relate_plan* p = d->plan;
  if (d->section >= p->W) return fail(RELATE_EINVAL, "no section left in this chunk");
  if (snp < 0 || snp >= d->L) return fail(RELATE_EINVAL, "SNP outside the chunk");
  // the interval of the section as its records carry it (src/anc_builder.cpp:66-67, src/fast_painting.cpp:603-606)
  const int start = p->wb[(size_t)d->section], end = p->wb[(size_t)d->section + 1] - 1;
  if (snp < start || snp > end) return fail(RELATE_EINVAL, "SNP outside the next section (the reference asserts, :106-107)");
  HIP_TRY(hipSetDevice(p->device));
  int64_t rows = 0;
  int pitch = 0;
  if (int rc = relate_plan_repaint_layout(p, d->section, &rows, &pitch, d->row_offsets.data())) return rc;
  if (int rc = d->topology.reserve(sizeof(float) * (size_t)rows * (size_t)pitch)) return rc;
  if (int rc = d->logscales.reserve(sizeof(float) * (size_t)rows)) return rc;
  if (d->ls_pending) {  // the previous section's copy may still be writing the block that is about to grow
    HIP_TRY(hipEventSynchronize(d->ls_fetched));
    d->ls_pending = false;
  }
  if (int rc = d->h_logscales.reserve(sizeof(float) * (size_t)rows)) return rc;
  const size_t R = relate_record_bytes(d->N);
  if (int rc = relate_plan_repaint(p, d->section, d->d_stones + (size_t)d->section * d->N * R, d->topology.p, d->logscales.p, stream))
    return rc;
  HIP_TRY(hipMemcpyAsync(d->h_logscales.p, d->logscales.p, sizeof(float) * (size_t)rows, hipMemcpyDeviceToHost, p->pick(stream)));
  HIP_TRY(hipEventRecord(d->ls_fetched, p->pick(stream)));
  d->ls_pending = true;
  d->rows = rows;
  d->pitch = pitch;
  d->section_startpos = start;
  d->section_endpos = end;
  // which row of its topology every haplotype stands at (:85-97) and where its last derived SNP lies (:100-105)
  std::fill(d->v_snp_prev.begin(), d->v_snp_prev.end(), 0);
  if (snp > 0)
    for (int t = snp; t >= start; t--)
      for (int n = 0; n < d->N; n++)
        if (d->is_derived(t, n)) d->v_snp_prev[(size_t)n]++;
  for (int n = 0; n < d->N; n++) {
    int t = snp;
    while (!d->is_derived(t, n) && t > 0) t--;
    d->v_rpos_prev[(size_t)n] = d->rpos[(size_t)t];
    d->v_rpos_next[(size_t)n] = d->v_rpos_prev[(size_t)n];
  }
  d->section++;
  return RELATE_OK;
}

int relate_distance_advance(relate_distance* d, int snp)
{
  if (!d || snp < 0 || snp >= d->L) return fail(RELATE_EINVAL, "bad argument");
  for (int n = 0; n < d->N; n++)
    if (d->is_derived(snp, n)) {
      d->v_snp_prev[(size_t)n]++;
      d->v_rpos_prev[(size_t)n] = d->rpos[(size_t)snp];
    }
  return RELATE_OK;
}

int relate_distance_get_matrix(relate_distance* d, int snp, void* d_matrix, void* stream)
{
  if (!d || !d_matrix) return fail(RELATE_EINVAL, "null argument");
  if (snp < 0 || snp >= d->L) return fail(RELATE_EINVAL, "SNP outside the chunk");
  if (snp > d->section_endpos)
    if (int rc = relate_distance_get_topology_with_repaint(d, snp, stream)) return rc;
  const int N = d->N, L = d->L;
  for (int n = 0; n < N; n++) {
    const bool direct = d->is_derived(snp, n) || snp == 0 || snp == L - 1;
    d->interpolate[(size_t)n] = direct ? 0 : 1;
    d->row_index[(size_t)n] = d->row_offsets[(size_t)n] + d->v_snp_prev[(size_t)n];
    d->weight_left[(size_t)n] = d->weight_right[(size_t)n] = 0.0;
    if (direct) continue;
    if (d->v_rpos_next[(size_t)n] <= d->v_rpos_prev[(size_t)n]) {  // :144-151
      for (int l = snp; l < L; l++)
        if (d->is_derived(l, n) || l == L - 1) {
          d->v_rpos_next[(size_t)n] = d->rpos[(size_t)l];
          break;
        }
    }
    const double rp = d->v_rpos_prev[(size_t)n], rn = d->v_rpos_next[(size_t)n];
    if (rp == rn) {
      d->weight_left[(size_t)n] = d->weight_right[(size_t)n] = 0.5;
    } else {
      const double denom = rn - rp;
      d->weight_left[(size_t)n] = (rn - d->rpos[(size_t)snp]) / denom;
      d->weight_right[(size_t)n] = (d->rpos[(size_t)snp] - rp) / denom;
    }
  }
  if (d->ls_pending) {
    HIP_TRY(hipSetDevice(d->plan->device));
    HIP_TRY(hipEventSynchronize(d->ls_fetched));
    d->ls_pending = false;
  }
  return plan_distance_matrix(d->plan, N, d->topology.p, d->pitch, d->logscales.p, d->h_logscales.as<float>(), d->rows,
                              d->row_index.data(), d->interpolate.data(), d->weight_left.data(), d->weight_right.data(),
                              d_matrix, stream);
}

int relate_distance_state(const relate_distance* d, int* section, int* section_startpos, int* section_endpos,
                          int64_t* v_snp_prev, double* v_rpos_prev, double* v_rpos_next)
{
  if (!d) return fail(RELATE_EINVAL, "null argument");
  if (section) *section = d->section;
  if (section_startpos) *section_startpos = d->section_startpos;
  if (section_endpos) *section_endpos = d->section_endpos;
  if (v_snp_prev) std::copy(d->v_snp_prev.begin(), d->v_snp_prev.end(), v_snp_prev);
  if (v_rpos_prev) std::copy(d->v_rpos_prev.begin(), d->v_rpos_prev.end(), v_rpos_prev);
  if (v_rpos_next) std::copy(d->v_rpos_next.begin(), d->v_rpos_next.end(), v_rpos_next);
  return RELATE_OK;
}

}  // extern "C"
