// TEST INFRASTRUCTURE ONLY: the MinMatch scan kernels (relate_amd/csrc/minmatch_kernels.hpp) run on CPU threads.
#include "hip_host_shim.hpp"

#include "../../relate_amd/csrc/minmatch_kernels.hpp"

#include <vector>

extern "C" long long hostrun_minmatch_scan(int N, const float* d, float threshold, float* min_values, long long capacity,
                                           int32_t* ci, int32_t* cj, float* cd)
{
  using namespace relate::minmatch;
  std::vector<int32_t> counts(N);
  std::vector<int64_t> offsets(N + 1);
  hostrun::launch(row_min_kernel, hostrun::Dim3(N), hostrun::Dim3(64), d, N, threshold, min_values);
  hostrun::launch(count_candidates_kernel, hostrun::Dim3(N), hostrun::Dim3(64), d, (const float*)min_values, N, counts.data());
  hostrun::launch(scan_counts_kernel, hostrun::Dim3(1), hostrun::Dim3(kScanThreads), (const int32_t*)counts.data(), N, offsets.data());
  hostrun::launch(fill_candidates_kernel, hostrun::Dim3(N), hostrun::Dim3(64), d, (const float*)min_values, N,
                  (const int64_t*)offsets.data(), (int64_t)capacity, ci, cj, cd);
  return offsets[N];
}
