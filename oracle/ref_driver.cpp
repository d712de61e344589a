// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin C-ABI driver around the reference's own painting translation unit
// (src/fast_painting.cpp, compiled from where it lies under /root/reference by
// oracle/Makefile; nothing of it is copied into this repository).  It fills the
// reference's `Data` struct in memory (all members are public, src/data.hpp:29-42)
// the way pipeline/Paint.cpp:32-58 and src/pipelines.rs:161-165 do, calls
// FastPainting::PaintSteppingStones / RePaintSection, and hands the bytes back.
// The built library lives in oracle/_ref/ (git-ignored).  It is used to pin
// oracle/paint_oracle.c and to generate tests/golden/, and may serve as the
// "reference" CPU baseline of the benchmark.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "data.hpp"
#include "fast_log.hpp"
#include "fast_painting.hpp"

namespace {

void fill_data(Data& data, int N, int L, const uint8_t* seq, const double* r, double theta_data)
{
  data.N = N;
  data.L = L;
  data.Ne = 30000;
  data.mu = 1.25e-8;
  data.name = "relate";
  data.theta = theta_data;
  data.ntheta = 1.0 - theta_data;
  data.sequence.resize(L, N);
  std::memcpy(&data.sequence[0][0], seq, (size_t)L * N);
  data.r.assign(r, r + L);
}

}  // namespace

extern "C" {

float ref_fast_log(float v) { return fast_log(v); }

// Same contract as oracle_paint_stones.  theta_painter == theta_data uses the
// C++ caller's constructor (pipeline/Paint.cpp:80), otherwise the Rust caller's
// (src/pipelines.rs:181).
int ref_paint_stones(int N, int L, const uint8_t* seq, const double* r, double theta_painter,
                     double theta_data, int W, const int32_t* wb, int k_begin, int k_end, uint8_t* out)
{
  Data data;
  fill_data(data, N, L, seq, r, theta_data);
  std::vector<int> window_boundaries(wb, wb + W + 1);
  std::vector<char*> bufs(W, nullptr);
  std::vector<size_t> sizes(W, 0);
  std::vector<FILE*> pfiles(W);
  for (int w = 0; w < W; w++) {
    pfiles[w] = open_memstream(&bufs[w], &sizes[w]);
    if (!pfiles[w]) return 1;
  }
  for (int k = k_begin; k < k_end; k++) {
    if (theta_painter == theta_data) {
      FastPainting painter(data);
      painter.PaintSteppingStones(data, window_boundaries, pfiles, k);
    } else {
      FastPainting painter((size_t)N, theta_painter);
      painter.PaintSteppingStones(data, window_boundaries, pfiles, k);
    }
  }
  const size_t R = 56 + (size_t)8 * N;
  const size_t K = (size_t)(k_end - k_begin);
  int rc = 0;
  for (int w = 0; w < W; w++) {
    fclose(pfiles[w]);
    if (sizes[w] != K * R) rc = 2;
    else std::memcpy(out + (size_t)w * K * R, bufs[w], K * R);
    free(bufs[w]);
  }
  return rc;
}

// The same with the reference's Data kept between calls (a handle): a caller that paints many small blocks of
// targets of one chunk, as the benchmark's CPU baseline does, fills the L x N matrix once instead of per call.
void* ref_data_new(int N, int L, const uint8_t* seq, const double* r, double theta_data)
{
  Data* data = new Data();
  fill_data(*data, N, L, seq, r, theta_data);
  return data;
}

void ref_data_free(void* h) { delete static_cast<Data*>(h); }

int ref_data_paint_stones(void* h, double theta_painter, int W, const int32_t* wb, int k_begin, int k_end, uint8_t* out)
{
  Data& data = *static_cast<Data*>(h);
  const int N = data.N;
  std::vector<int> window_boundaries(wb, wb + W + 1);
  std::vector<char*> bufs(W, nullptr);
  std::vector<size_t> sizes(W, 0);
  std::vector<FILE*> pfiles(W);
  for (int w = 0; w < W; w++) {
    pfiles[w] = open_memstream(&bufs[w], &sizes[w]);
    if (!pfiles[w]) return 1;
  }
  for (int k = k_begin; k < k_end; k++) {
    if (theta_painter == data.theta) {
      FastPainting painter(data);
      painter.PaintSteppingStones(data, window_boundaries, pfiles, k);
    } else {
      FastPainting painter((size_t)N, theta_painter);
      painter.PaintSteppingStones(data, window_boundaries, pfiles, k);
    }
  }
  const size_t R = 56 + (size_t)8 * N;
  const size_t K = (size_t)(k_end - k_begin);
  int rc = 0;
  for (int w = 0; w < W; w++) {
    fclose(pfiles[w]);
    if (sizes[w] != K * R) rc = 2;
    else std::memcpy(out + (size_t)w * K * R, bufs[w], K * R);
    free(bufs[w]);
  }
  return rc;
}

int ref_repaint_section(int N, int L, const uint8_t* seq, const double* r, double theta, int first, int last,
                        const float* alpha_begin, const float* beta_end, float logscale_alpha,
                        float logscale_beta, int k, float* topology, float* logscales)
{
  Data data;
  fill_data(data, N, L, seq, r, theta);
  CollapsedMatrix<float> top, ab, be;
  ab.resize(1, N);
  be.resize(1, N);
  std::memcpy(ab[0], alpha_begin, sizeof(float) * N);
  std::memcpy(be[0], beta_end, sizeof(float) * N);
  std::vector<float> ls;
  FastPainting painter(data);
  painter.RePaintSection(data, top, ls, ab, be, first, last, logscale_alpha, logscale_beta, k);
  const int nd = (int)top.size();
  std::memcpy(topology, top[0], sizeof(float) * (size_t)nd * N);
  std::memcpy(logscales, ls.data(), sizeof(float) * nd);
  return nd;
}

}  // extern "C"
