// treebuilder_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the tree builder that consumes the distance matrix, MinMatch::QuickBuild of tcztzy/relate
// (src/tree_builder.cpp:8-23 comparison of candidates, :57-127 and :130-214 Initialize, :217-255 InitializeSym,
// :258-475 and :478-752 Coalesce, :755-845 CoalesceSym, :848-1064 QuickBuild; types from src/tree_builder.hpp:12-86).
// The library builds only the two scans of Initialize on the GPU; the rest is restated here so that the scans can be
// checked where the reference itself holds known answers: the parent arrays of test/test_treebuilder.cpp:36-86 and
// :88-137 come out of QuickBuild, not out of Initialize alone.
//
// Parity status: the reference's tree_builder.cpp cannot be compiled in this image (tree_builder.hpp -> anc.hpp ->
// <gzstream.h>, a library the image lacks); this restatement is pinned by those two known-answer trees and by
// nothing else: PARITY UNPINNED beyond them.  Random numbers: std::mt19937 seeded with 1 and
// std::uniform_real_distribution<double>(0,1) of the C++ library, as the reference (:852-853); the draws are
// narrowed to float where the reference stores them in its float member dist_random (src/tree_builder.hpp:47).
//
// Build: g++ -O2 -ffp-contract=off, linked into libpaint_oracle.so.  Only tests/ may load it.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <limits>
#include <random>
#include <vector>

namespace {

const float kInf = std::numeric_limits<float>::infinity();

struct Cand {
  int lin1 = -1, lin2 = -1;
  double dist = kInf, dist2 = kInf, dist3 = kInf;
  bool replace = false;
};

// bool operator>(const Candidate& a, const Candidate& b)  (:8-23)
bool greater(const Cand& a, const Cand& b)
{
  if (a.replace == true && a.dist3 >= b.dist3) {
    if (a.dist3 > b.dist3) return true;
    if (a.dist > b.dist || (a.dist == b.dist && a.dist2 > b.dist2)) return true;
  }
  if (a.dist > b.dist || (a.dist == b.dist && a.dist2 > b.dist2)) return true;
  return false;
}

struct Builder {
  int N = 0, Ne = 30000;
  float threshold = 0.f;
  bool with_ages = false;
  double age = 0.0;
  std::mt19937 rng;
  std::uniform_real_distribution<double> unif{0.0, 1.0};
  std::vector<float> d;  // N x N, row-major, updated in place as clusters merge
  std::vector<float> sym;
  std::vector<int> convert_index, updated;
  std::vector<float> cluster_size, min_values, min_values_sym;
  std::deque<int> cluster;
  std::vector<Cand> m, msym;
  Cand best, best_sym;
  std::vector<double> ages;
  std::vector<double> unique_ages;
  std::vector<int> ages_count;

  float& D(int a, int b) { return d[(size_t)a * N + b]; }
  float& S(int a, int b) { return sym[(size_t)a * N + b]; }

  // One pair (a, b) that passed both row tests: its symmetric distance, one draw, and the two offers in the
  // order the call site makes them.  lin1/lin2 are what the reference writes into the slot.
  void offer_pair(float sym_dist, int first, int second, int lin1, int lin2)
  {
    if (!with_ages) {
      const float dist_random = (float)unif(rng);  // float member, src/tree_builder.hpp:47
      for (int slot : {first, second}) {
        Cand& c = m[slot];
        if (c.dist > sym_dist || (c.dist == sym_dist && c.dist2 > dist_random)) {
          c.lin1 = lin1;
          c.lin2 = lin2;
          c.dist = sym_dist;
          c.dist2 = dist_random;
        }
      }
    } else {
      Cand cand;
      cand.dist = sym_dist;
      cand.dist3 = std::max(ages[lin1], ages[lin2]);
      cand.dist2 = unif(rng);
      for (int slot : {first, second}) {
        Cand& c = m[slot];
        if ((c.dist == kInf || cand.dist3 <= age) && greater(c, cand)) {
          cand.replace = cand.dist3 > age;
          c = cand;
          c.lin1 = lin1;
          c.lin2 = lin2;
        }
      }
    }
  }
  void reset(Cand& c)
  {
    c.dist = kInf;
    c.dist2 = kInf;
    if (with_ages) {
      c.dist3 = kInf;
      c.replace = false;
    }
  }
  // best_candidate against mcandidates[k] after a coalescence (:428-430, :682-689, :471-473, :743-750)
  void best_against(const Cand& c)
  {
    if (!with_ages) {
      if (best.dist > c.dist || (best.dist == c.dist && best.dist2 > c.dist2)) best = c;
    } else if ((best.dist == kInf || c.dist3 <= age) && greater(best, c)) {
      best = c;
      best.replace = best.dist3 > age;
    }
  }

  // :57-127 / :130-214
  void initialize()
  {
    size_t pos = 0;
    for (int it : cluster) {
      reset(m[it]);
      float& mv = min_values[pos];
      for (int l : cluster)
        if (mv > D(it, l) && l != it) mv = D(it, l);
      mv += threshold;
      pos++;
    }
    // min_values is walked by position, the matrix by cluster index: the same thing while every leaf is its own
    // cluster (QuickBuild fills cluster_index with 0 .. N-1, :866-883)
    for (size_t a = 0; a < cluster.size(); a++) {
      const int it = cluster[a];
      for (size_t b = a + 1; b < cluster.size(); b++) {
        const int jt = cluster[b];
        if (min_values[a] >= D(it, jt)) {
          if (min_values[b] >= D(jt, it)) {
            const float sym_dist = D(it, jt) + D(jt, it);
            if (!with_ages) {
              offer_pair(sym_dist, it, jt, it, jt);
              const Cand& cj = m[jt];
              if (best.dist > cj.dist || (best.dist == cj.dist && best.dist2 > cj.dist2)) {  // :109-114
                best.lin1 = it;
                best.lin2 = jt;
                best.dist = sym_dist;
                best.dist2 = cj.dist2;
              }
            } else {
              offer_pair(sym_dist, it, jt, it, jt);
              best_against(m[jt]);  // :193-200
            }
          }
        }
      }
    }
  }

  // :217-255
  void initialize_sym()
  {
    sym.assign((size_t)N * N, 0.f);
    for (size_t a = 0; a < cluster.size(); a++)
      for (size_t b = a + 1; b < cluster.size(); b++) {
        const int it = cluster[a], jt = cluster[b];
        S(it, jt) = D(it, jt) + D(jt, it);
        S(jt, it) = S(it, jt);
      }
    for (int it : cluster) {
      float& mv = min_values_sym[it];
      msym[it].dist = kInf;
      for (int l : cluster) {
        if (mv > S(it, l) && l != it) {
          mv = S(it, l);
          if (msym[it].dist > mv) {
            msym[it].lin1 = it;
            msym[it].lin2 = l;
            msym[it].dist = mv;
          }
          if (best_sym.dist > msym[it].dist) {
            best_sym.lin1 = it;
            best_sym.lin2 = l;
            best_sym.dist = mv;
          }
        }
      }
    }
  }

  // :258-475 / :478-752
  void coalesce(int i, int j)
  {
    const float added = cluster_size[i] + cluster_size[j];
    float min_value_j = kInf;
    int nupd = 0;
    best.dist = kInf;
    best.dist2 = kInf;
    if (with_ages) {
      best.dist3 = kInf;
      best.replace = false;
    }
    for (size_t kp = 0; kp < cluster.size(); kp++) {
      const int k = cluster[kp];
      if (k == j || k == i) continue;
      const float dkj = D(k, j), dki = D(k, i), dik = D(i, k), djk = D(j, k);
      float mvk = min_values[k];
      if (with_ages && m[k].dist3 <= age) m[k].replace = false;
      if (dik != djk) D(j, k) = (cluster_size[i] * dik + cluster_size[j] * djk) / added;
      if (dki != dkj) D(k, j) = (cluster_size[i] * dki + cluster_size[j] * dkj) / added;
      bool changed = false;
      if (dkj != dki) {
        if (std::fabs(mvk - threshold - dkj) < 1e-4 || std::fabs(mvk - threshold - dki) < 1e-4) {
          const float old = mvk - threshold;
          mvk = kInf;
          changed = true;
          for (int l : cluster)
            if (l != i && l != k)
              if (mvk > D(k, l)) {
                mvk = D(k, l);
                if (mvk == old) break;
              }
          mvk += threshold;
          min_values[k] = mvk;
        }
      }
      // pairs of k with the clusters whose candidates were reset earlier in this sweep
      auto against_updated = [&]() {
        for (int u = 0; u < nupd; u++) {
          const int l = updated[u];
          if (D(k, l) <= mvk)
            if (D(l, k) <= min_values[l]) offer_pair(D(l, k) + D(k, l), l, k, k, l);
        }
      };
      if (dkj != dki || djk != dik) {
        if (changed || m[k].lin1 == j || m[k].lin2 == j || m[k].lin1 == i || m[k].lin2 == i) {
          updated[nupd++] = k;
          reset(m[k]);
          for (size_t lp = 0; lp < kp; lp++) {
            const int l = cluster[lp];
            if (D(k, l) <= mvk) {
              const float mvl = min_values[l];
              if (l != j && l != i)
                if (D(l, k) <= mvl) offer_pair(D(l, k) + D(k, l), k, l, k, l);
            }
          }
        } else {
          against_updated();
        }
      } else {
        if (m[k].lin1 == i) m[k].lin1 = j;
        if (m[k].lin2 == i) m[k].lin2 = j;
        against_updated();
      }
      best_against(m[k]);
      if (D(j, k) < min_value_j) min_value_j = D(j, k);
    }
    min_value_j += threshold;
    min_values[j] = min_value_j;
    reset(m[j]);
    for (int k : cluster)
      if (D(j, k) <= min_value_j)
        if (D(k, j) <= min_values[k])
          if (k != i && k != j) offer_pair(D(j, k) + D(k, j), k, j, k, j);
    best_against(m[j]);
  }

  // :755-845
  void coalesce_sym(int i, int j)
  {
    const float added = cluster_size[i] + cluster_size[j];
    float min_value_j = kInf;
    best_sym.dist = kInf;
    msym[j].dist = kInf;
    for (int k : cluster) {
      if (k == j || k == i) continue;
      const float dkj = S(k, j), dki = S(k, i), dik = S(i, k), djk = S(j, k);
      float mvk = min_values_sym[k];
      if (dik != djk) S(j, k) = (cluster_size[i] * dik + cluster_size[j] * djk) / added;
      if (dki != dkj) S(k, j) = (cluster_size[i] * dki + cluster_size[j] * dkj) / added;
      if (dkj != dki) {
        if (std::fabs(mvk - dkj) < 1e-6 || std::fabs(mvk - dki) < 1e-6) {
          const float old = mvk;
          mvk = kInf;
          msym[k].dist = kInf;
          for (int l : cluster)
            if (l != i && l != k)
              if (mvk > S(k, l)) {
                mvk = S(k, l);
                if (msym[k].dist > mvk) {
                  msym[k].lin1 = k;
                  msym[k].lin2 = l;
                  msym[k].dist = mvk;
                }
                if (mvk == old) break;
              }
          min_values_sym[k] = mvk;
        }
      } else {
        if (msym[k].lin1 == i) msym[k].lin1 = j;
        if (msym[k].lin2 == i) msym[k].lin2 = j;
      }
      if (best_sym.dist > msym[k].dist) best_sym = msym[k];
      if (S(j, k) < min_value_j) {
        min_value_j = S(j, k);
        if (msym[j].dist > S(j, k)) {
          msym[j].lin1 = k;
          msym[j].lin2 = j;
          msym[j].dist = S(j, k);
        }
      }
    }
    min_values_sym[j] = min_value_j;
    if (best_sym.dist > msym[j].dist) best_sym = msym[j];
  }

  // MinMatch::MinMatch (:38-54) and the set-up of QuickBuild (:852-892)
  void setup(int n, const float* matrix, double theta, int ne, const double* sample_ages)
  {
    N = n;
    Ne = ne;
    threshold = (float)(-0.2 * std::log(theta / (1.0 - theta)));
    d.assign(matrix, matrix + (size_t)n * n);
    rng.seed(1);
    unif = std::uniform_real_distribution<double>(0.0, 1.0);
    convert_index.resize(n);
    cluster_size.assign(n, 1.0f);
    updated.assign(n, 0);
    cluster.resize(n);
    for (int c = 0; c < n; c++) cluster[c] = convert_index[c] = c;
    min_values.assign(n, kInf);
    min_values_sym.assign(n, kInf);
    m.assign(n, Cand());
    msym.assign(n, Cand());
    best = Cand();
    best_sym = Cand();
    with_ages = sample_ages != nullptr;
    if (with_ages) ages.assign(sample_ages, sample_ages + n);
  }

  // :894-999 / :1001-1059; parents[c] = label of the parent of node c, -1 for the root
  int quick_build(int32_t* parents)
  {
    const int total = 2 * N - 1;
    std::fill(parents, parents + total, -1);
    int level = 0, num_lins = 0;
    if (with_ages) {
      std::vector<double> sorted = ages;
      std::sort(sorted.begin(), sorted.end());
      unique_ages.assign(sorted.size(), 0.0);
      ages_count.assign(sorted.size(), 0);
      age = sorted[0];
      int u = 0;
      unique_ages[0] = age;
      for (double a : sorted) {
        if (a == age) {
          ages_count[u]++;
        } else {
          age = a;
          u++;
          unique_ages[u] = age;
          ages_count[u]++;
        }
      }
      unique_ages.resize(u + 1);
      ages_count.resize(u + 1);
      num_lins = ages_count[0];
      age = unique_ages[0] + 2.0 / ((double)num_lins * (num_lins - 1.0)) * Ne;
    }
    initialize();
    bool use_sym = false;
    int fallbacks = 0;
    for (int node = N; node < total; node++) {
      int i, j;
      if (best.dist == kInf) {
        fallbacks++;
        if (!use_sym) {
          initialize_sym();
          use_sym = true;
        }
        if (best_sym.dist == kInf) return -1;  // the reference asserts
        i = best_sym.lin1;
        j = best_sym.lin2;
      } else {
        i = best.lin1;
        j = best.lin2;
      }
      if (i == j) return -1;
      parents[convert_index[i]] = node;
      parents[convert_index[j]] = node;
      coalesce(i, j);
      if (use_sym) coalesce_sym(i, j);
      if (with_ages) {
        ages[j] = std::max(ages[i], ages[j]);
        num_lins--;
        while (unique_ages[level] < ages[j]) {
          level++;
          num_lins += ages_count[level];
        }
        age += 2.0 / ((double)num_lins * (num_lins - 1.0)) * Ne;
      }
      cluster_size[j] = cluster_size[i] + cluster_size[j];
      convert_index[j] = node;
      for (auto it = cluster.begin(); it != cluster.end(); ++it)
        if (*it == i) {
          cluster.erase(it);
          break;
        }
    }
    return fallbacks;
  }
};

}  // namespace

extern "C" {

// threshold of MinMatch for Data::theta (:43)
float oracle_minmatch_threshold(double theta) { return (float)(-0.2 * std::log(theta / (1.0 - theta))); }

// The two scans of Initialize alone: min_values (threshold added) and the candidate pairs in the order of the
// reference's double loop.  Returns the number of pairs (which may exceed capacity; only the first `capacity`
// are stored).
long long oracle_minmatch_scan(int N, const float* d, float threshold, float* min_values, long long capacity, int32_t* ci,
                               int32_t* cj, float* cd)
{
  for (int i = 0; i < N; i++) {
    float mv = kInf;
    for (int l = 0; l < N; l++)
      if (mv > d[(size_t)i * N + l] && l != i) mv = d[(size_t)i * N + l];
    min_values[i] = mv + threshold;
  }
  long long n = 0;
  for (int i = 0; i < N; i++)
    for (int j = i + 1; j < N; j++)
      if (min_values[i] >= d[(size_t)i * N + j])
        if (min_values[j] >= d[(size_t)j * N + i]) {
          if (n < capacity) {
            ci[n] = i;
            cj[n] = j;
            cd[n] = d[(size_t)i * N + j] + d[(size_t)j * N + i];
          }
          n++;
        }
  return n;
}

// Initialize as a whole (generator seeded with 1 as in QuickBuild): the candidates of every cluster and the
// best candidate.  sample_ages == NULL: :57-127; otherwise :130-214 with the first coalescence age of QuickBuild
// (:924-926).  m_*: N entries each; best: lin1, lin2 in best_lin, dist / dist2 / dist3 in best_dist.
void oracle_minmatch_initialize(int N, const float* d, double theta, int Ne, const double* sample_ages, int32_t* m_lin1,
                                int32_t* m_lin2, double* m_dist, double* m_dist2, int32_t* best_lin, double* best_dist,
                                float* min_values)
{
  Builder b;
  b.setup(N, d, theta, Ne, sample_ages);
  if (sample_ages) {
    std::vector<double> s(sample_ages, sample_ages + N);
    std::sort(s.begin(), s.end());
    int cnt = 0;
    for (double a : s) cnt += a == s[0];
    b.age = s[0] + 2.0 / ((double)cnt * (cnt - 1.0)) * Ne;
  }
  b.initialize();
  for (int c = 0; c < N; c++) {
    m_lin1[c] = b.m[c].lin1;
    m_lin2[c] = b.m[c].lin2;
    m_dist[c] = b.m[c].dist;
    m_dist2[c] = b.m[c].dist2;
    min_values[c] = b.min_values[c];
  }
  best_lin[0] = b.best.lin1;
  best_lin[1] = b.best.lin2;
  best_dist[0] = b.best.dist;
  best_dist[1] = b.best.dist2;
  best_dist[2] = b.best.dist3;
}

// MinMatch::QuickBuild: parents of the 2N-1 nodes (leaves 0 .. N-1, merged nodes N .. 2N-2 in the order of the
// coalescences, root -1).  Returns the number of merges that fell back to the symmetric matrix, -1 on failure.
int oracle_quick_build(int N, const float* d, double theta, int Ne, const double* sample_ages, int32_t* parents)
{
  Builder b;
  b.setup(N, d, theta, Ne, sample_ages);
  return b.quick_build(parents);
}

// n draws of uniform_real_distribution<double>(0,1) from mt19937 seeded with `seed`: what the reference's
// generator hands out, in order.
void oracle_uniform_draws(unsigned seed, int n, double* out)
{
  std::mt19937 rng;
  rng.seed(seed);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  for (int i = 0; i < n; i++) out[i] = u(rng);
}

}  // extern "C"
