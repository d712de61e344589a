// TEST INFRASTRUCTURE ONLY.  Runs plain-HIP kernel source on CPU threads, so that a kernel's text can be checked
// against its oracle on a machine without a GPU: one std::thread per work-item of a workgroup, workgroups one after
// the other; __syncthreads is a barrier over the workgroup, the wave operations (__ballot, __shfl*, 64 lanes) go
// through a slot array and a barrier per wave.  Only what the library's plain kernels use is provided.  A kernel
// file is compiled with -DRELATE_HOST_RUN -include this header (g++ -std=c++20).
#pragma once
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace hostrun {
struct Dim3 {
  unsigned x = 1, y = 1, z = 1;
  Dim3() {}
  Dim3(unsigned x_, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct Block {
  std::barrier<> all;
  std::vector<std::unique_ptr<std::barrier<>>> wave;
  std::vector<uint64_t> slot;
  explicit Block(unsigned n) : all((std::ptrdiff_t)n), slot(n)
  {
    for (unsigned w = 0; w * 64 < n; w++) wave.emplace_back(new std::barrier<>((std::ptrdiff_t)(n - w * 64 < 64 ? n - w * 64 : 64)));
  }
};
inline thread_local Block* g_block = nullptr;
inline thread_local unsigned g_tid = 0;
}  // namespace hostrun

inline thread_local hostrun::Dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

inline void __syncthreads() { hostrun::g_block->all.arrive_and_wait(); }

namespace hostrun {
// every lane of the wave deposits `bits`, then reads the slot of lane `src` (of its own wave)
inline uint64_t exchange(uint64_t bits, unsigned src_lane)
{
  Block& b = *g_block;
  const unsigned w = g_tid >> 6;
  b.slot[g_tid] = bits;
  b.wave[w]->arrive_and_wait();
  const unsigned idx = (w << 6) | (src_lane & 63);
  const uint64_t r = idx < b.slot.size() ? b.slot[idx] : bits;
  b.wave[w]->arrive_and_wait();
  return r;
}
template <class T>
inline uint64_t to_bits(T v)
{
  uint64_t x = 0;
  static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
  std::memcpy(&x, &v, sizeof(T));
  return x;
}
template <class T>
inline T from_bits(uint64_t x)
{
  T v;
  std::memcpy(&v, &x, sizeof(T));
  return v;
}
}  // namespace hostrun

template <class T>
inline T __shfl_xor(T v, int mask, int width = 64)
{
  (void)width;
  return hostrun::from_bits<T>(hostrun::exchange(hostrun::to_bits(v), (hostrun::g_tid & 63) ^ (unsigned)mask));
}
template <class T>
inline T __shfl(T v, int lane, int width = 64)
{
  (void)width;
  return hostrun::from_bits<T>(hostrun::exchange(hostrun::to_bits(v), (unsigned)lane));
}
template <class T>
inline T __shfl_up(T v, unsigned delta, int width = 64)
{
  (void)width;
  const unsigned lane = hostrun::g_tid & 63;
  return hostrun::from_bits<T>(hostrun::exchange(hostrun::to_bits(v), lane >= delta ? lane - delta : lane));
}
inline unsigned long long __ballot(int pred)
{
  // 64 exchanges would do; one is enough: every lane publishes its bit, then all read all
  hostrun::Block& b = *hostrun::g_block;
  const unsigned w = hostrun::g_tid >> 6;
  b.slot[hostrun::g_tid] = pred ? 1 : 0;
  b.wave[w]->arrive_and_wait();
  unsigned long long m = 0;
  for (unsigned l = 0; l < 64 && (w << 6) + l < b.slot.size(); l++)
    if (b.slot[(w << 6) + l]) m |= 1ull << l;
  b.wave[w]->arrive_and_wait();
  return m;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }

namespace hostrun {
// kernel<<<grid, block>>>(args...) on the host
template <class K, class... A>
inline void launch(K kernel, Dim3 grid, Dim3 block, A... args)
{
  const unsigned n = block.x * block.y * block.z;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        Block ctx(n);
        std::vector<std::thread> th;
        th.reserve(n);
        for (unsigned t = 0; t < n; t++)
          th.emplace_back([&, t] {
            g_block = &ctx;
            g_tid = t;
            threadIdx = Dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            blockIdx = Dim3(bx, by, bz);
            blockDim = block;
            gridDim = grid;
            kernel(args...);
            // a work-item that has returned no longer takes part in barriers
            ctx.wave[t >> 6]->arrive_and_drop();
            ctx.all.arrive_and_drop();
          });
        for (auto& x : th) x.join();
      }
}
}  // namespace hostrun
