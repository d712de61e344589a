// Seeded synthetic chunks (host-only).  The reference ships no simulator and
// msprime is not available here, so the benchmark and the large parity cases
// draw their inputs from this generator: a coalescent tree over the N
// haplotypes that changes along the chunk by prune-and-regraft moves (so
// neighbouring SNPs are in linkage disequilibrium, which decays with distance),
// one mutation per SNP placed on a branch chosen in proportion to its length
// (which yields the 1/i site-frequency spectrum), and a recombination map that
// is flat (about 1 cM/Mb at 1 kb spacing, r ~ 0.025 in the units of
// src/data.cpp:481-491) or has hotspots (log-uniform in [2.5e-7, 10]).
#include "../../include/relate_gpu.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "chunk_io.hpp"

namespace {

struct Tree {
  int N;
  std::vector<int> parent, c0, c1;
  std::vector<double> time;
  int root;
  explicit Tree(int n) : N(n), parent(2 * n - 1, -1), c0(2 * n - 1, -1), c1(2 * n - 1, -1), time(2 * n - 1, 0.0), root(-1) {}

  void build(std::mt19937_64& rng)
  {
    std::vector<int> live(N);
    for (int i = 0; i < N; i++) live[i] = i;
    double t = 0.0;
    int next = N;
    while (live.size() > 1) {
      const double k = (double)live.size();
      std::exponential_distribution<double> wait(k * (k - 1.0) / 2.0);
      t += wait(rng);
      std::uniform_int_distribution<size_t> pick(0, live.size() - 1);
      size_t i = pick(rng), j = pick(rng);
      while (j == i) j = pick(rng);
      const int a = live[i], b = live[j], p = next++;
      c0[p] = a;
      c1[p] = b;
      parent[a] = parent[b] = p;
      time[p] = t;
      if (i < j) std::swap(i, j);
      live.erase(live.begin() + (long)i);
      live.erase(live.begin() + (long)j);
      live.push_back(p);
    }
    root = live[0];
  }

  void replace_child(int p, int from, int to)
  {
    if (c0[p] == from) c0[p] = to;
    else c1[p] = to;
  }

  std::vector<double> sorted_times;  // times of the internal nodes, ascending

  void init_sorted()
  {
    sorted_times.assign(time.begin() + N, time.end());
    std::sort(sorted_times.begin(), sorted_times.end());
  }

  // One recombination event of the sequentially Markov coalescent (SMC'): the cut point is uniform on
  // the total branch length (branch x, height h); the floating lineage then coalesces with the k(t)
  // lineages of the remaining tree at rate k(t), with one of them chosen uniformly.
  // cum[] = running sum of branch lengths.  Returns false if the draw leaves the tree unchanged.
  bool spr(std::mt19937_64& rng, const std::vector<double>& cum, std::vector<int>& stack, std::vector<int>& mark,
           int stamp)
  {
    const int nn = 2 * N - 1;
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const double xr = U(rng) * cum[nn - 1];
    int x = (int)(std::lower_bound(cum.begin(), cum.end(), xr) - cum.begin());
    if (x >= nn) x = nn - 1;
    if (x == root) return false;
    const int p = parent[x];
    const double h = time[x] + U(rng) * (time[p] - time[x]);
    const double tp_old = time[p];
    // coalescence time of the floating lineage: walk the intervals between node times above h
    // (p itself is gone from the remaining tree)
    size_t idx = (size_t)(std::upper_bound(sorted_times.begin(), sorted_times.end(), h) - sorted_times.begin());
    double t = h, tnew = -1.0;
    for (;;) {
      // internal nodes with time <= t, not counting p
      size_t below = idx;
      if (tp_old <= t) below -= 1;
      const double k = (double)N - 1.0 - (double)below;  // lineages of the remaining tree at time t
      double next = idx < sorted_times.size() ? sorted_times[idx] : -1.0;
      std::exponential_distribution<double> wait(k < 1.0 ? 1.0 : k);
      const double w = wait(rng);
      if (next < 0.0 || t + w < next) {
        tnew = t + w;
        break;
      }
      t = next;
      idx++;
    }
    // mark the subtree of x, then draw a lineage alive at tnew uniformly (rejection over all nodes)
    stack.clear();
    stack.push_back(x);
    while (!stack.empty()) {
      const int v = stack.back();
      stack.pop_back();
      mark[v] = stamp;
      if (c0[v] >= 0) {
        stack.push_back(c0[v]);
        stack.push_back(c1[v]);
      }
    }
    const int sib = (c0[p] == x) ? c1[p] : c0[p];
    const int g = parent[p];
    std::uniform_int_distribution<int> pick(0, nn - 1);
    int z = -1;
    for (int tries = 0; tries < 64 * nn && z < 0; tries++) {
      const int cand = pick(rng);
      if (cand == p || mark[cand] == stamp || time[cand] > tnew) continue;
      // parent in the remaining tree (sib hangs under g once p is removed)
      const int cp = (cand == sib) ? g : parent[cand];
      if (cp >= 0 && time[cp] <= tnew) continue;
      z = cand;
    }
    if (z < 0) return false;
    if (z == sib && tnew == tp_old) return false;
    // detach: sib takes p's place
    parent[sib] = g;
    if (g >= 0) replace_child(g, p, sib);
    else root = sib;
    // attach p above z
    const int zp = parent[z];
    if (zp >= 0) replace_child(zp, z, p);
    else root = p;
    parent[p] = zp;
    c0[p] = x;
    c1[p] = z;
    parent[x] = p;
    parent[z] = p;
    time[p] = tnew;
    sorted_times.erase(std::lower_bound(sorted_times.begin(), sorted_times.end(), tp_old));
    sorted_times.insert(std::upper_bound(sorted_times.begin(), sorted_times.end(), tnew), tnew);
    return true;
  }
};

}  // namespace

extern "C" int relate_synth_chunk(int N, int L, uint64_t seed, int skewed_map, uint8_t* seq, double* r)
{
  if (N < 2 || L < 2 || !seq || !r) return RELATE_EINVAL;
  std::mt19937_64 rng(seed * 0x9E3779B97F4A7C15ull + 12345u);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  for (int s = 0; s < L; s++) {
    if (skewed_map) r[s] = std::exp(std::log(2.5e-7) + U(rng) * (std::log(10.0) - std::log(2.5e-7)));
    else r[s] = 0.025 * (0.5 + U(rng));
  }
  Tree tr(N);
  tr.build(rng);
  tr.init_sorted();
  const int nn = 2 * N - 1;
  std::vector<int> stack, mark(nn, 0);
  std::vector<double> cum(nn);
  int stamp = 0;
  bool dirty = true;
  std::memset(seq, '0', (size_t)N * L);
  for (int s = 0; s < L; s++) {
    int moves = (int)std::floor(20.0 * r[s]);
    if (U(rng) < 20.0 * r[s] - moves) moves++;
    if (moves > 8) moves = 8;
    auto refresh = [&]() {
      if (!dirty) return;
      double acc = 0.0;
      for (int v = 0; v < nn; v++) {
        if (v != tr.root) acc += tr.time[tr.parent[v]] - tr.time[v];
        cum[v] = acc;
      }
      dirty = false;
    };
    for (int m = 0; m < moves; m++) {
      refresh();
      if (tr.spr(rng, cum, stack, mark, ++stamp)) dirty = true;
    }
    refresh();
    const double x = U(rng) * cum[nn - 1];
    int v = (int)(std::lower_bound(cum.begin(), cum.end(), x) - cum.begin());
    if (v >= nn) v = nn - 1;
    if (v == tr.root) v = tr.c0[v];
    uint8_t* row = seq + (size_t)s * N;
    stack.clear();
    stack.push_back(v);
    while (!stack.empty()) {
      const int u = stack.back();
      stack.pop_back();
      if (u < N) row[u] = '1';
      else {
        stack.push_back(tr.c0[u]);
        stack.push_back(tr.c1[u]);
      }
    }
  }
  return RELATE_OK;
}

extern "C" int relate_window_boundaries(int N, int L, const uint8_t* seq, double memory_gb, int max_windows, int32_t* wb)
{
  // src/data.cpp:145,:219-255
  if (N < 2 || L < 2 || !seq || !wb || max_windows < 1) return RELATE_EINVAL;
  const double budget = memory_gb * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N);
  if (budget <= 0) return RELATE_ERANGE;  // "Need larger memory allowance"
  int nw = 1, in_window = 0;
  double mem = 0.0;
  wb[0] = 0;
  for (int s = 0; s < L; s++) {
    int derived = 0;
    const uint8_t* row = seq + (size_t)s * N;
    for (int n = 0; n < N; n++) derived += row[n] == '1';
    mem += (double)derived * (N + 1);
    if (mem >= budget && in_window > 10 && nw < max_windows) {
      in_window = 0;
      mem = 0.0;
      wb[nw++] = s;
    }
    in_window++;
  }
  wb[nw] = L;
  return nw;
}

extern "C" int relate_write_chunk(const char* output_dir, int chunk_index, int N, int L, const uint8_t* seq,
                                  const double* r, int W, const int32_t* wb)
{
  if (!output_dir || !seq || !r || !wb || W < 1) return RELATE_EINVAL;
  using namespace relate::io;
  const std::string base = std::string(output_dir) + "/chunk_" + std::to_string(chunk_index);
  std::string err;
  std::vector<int32_t> wbv(wb, wb + W + 1), bp(L), dist(L), state(L, 1);
  std::vector<double> rv(r, r + L), rpos(L + 1);
  rpos[0] = 0.0;
  for (int s = 0; s < L; s++) {
    bp[s] = 1000 * (s + 1);
    dist[s] = 1000;
    rpos[s + 1] = rpos[s] + r[s] / 2500.0;  // inverse of src/data.cpp:484-488
  }
  const bool ok = write_parameters(std::string(output_dir) + "/parameters_c" + std::to_string(chunk_index) + ".bin", N, L,
                                   wbv, err) &&
                  write_hap(base + ".hap", N, L, seq, err) && write_vector(base + ".bp", bp, err) &&
                  write_vector(base + ".dist", dist, err) && write_vector(base + ".r", rv, err) &&
                  write_vector(base + ".rpos", rpos, err) && write_state(base + ".state", state, err);
  return ok ? RELATE_OK : RELATE_EIO;
}
