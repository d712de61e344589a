/*
 * paint_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, double-precision CPU restatement of the Li-Stephens stepping-stone
 * painting of tcztzy/relate, written from the algorithm (not from its text) so
 * that the GPU path has an independent checker.  Only tests/, the smoke check
 * and the cpu_baseline leg of the benchmark may load this file; the product
 * library never does.
 *
 * Parity status: PINNED.  Every function here is checked byte-for-byte against
 * the reference's own fast_painting.cpp compiled in this container
 * (oracle/_ref, see oracle/Makefile) on the seeded cases under tests/golden/,
 * and against the known-answer test of the reference (test/test_painting.cpp:8-136,
 * test/test_log.cpp:6-15).
 *
 * Build: gcc -O2 -ffp-contract=off (no -march, no -ffast-math): fast_log's
 * polynomial must round after every float operation, as the reference build does.
 *
 * Reference citations are relative to the reference tree (src/..., pipeline/...).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- fast_log
 * src/fast_log.hpp:6-21.  Exponent taken from the float's bits (biased by 128,
 * not 127), mantissa forced into [1,2), quadratic polynomial, then ln 2. */
float oracle_fast_log(float val)
{
  int32_t x;
  memcpy(&x, &val, 4);
  const int log_2 = ((x >> 23) & 255) - 128;
  x &= ~(255 << 23);
  x += 127 << 23;
  memcpy(&val, &x, 4);
  val = ((-1.0f / 3) * val + 2) * val - 2.0f / 3;
  return (val + log_2) * 0.69314718f;
}

/* Constants of one painter: src/fast_painting.hpp:19-45 (both constructors
 * produce these values from theta and N). */
typedef struct {
  double Nm1, prior_theta, prior_ntheta, theta_ratio, log_ntheta, log_small, lo, hi;
} painter_t;

static painter_t make_painter(int N, double theta)
{
  painter_t p;
  double ntheta = 1.0 - theta;
  p.lo = 1e-10;
  p.hi = 1.0 / p.lo;
  p.Nm1 = (double)N - 1.0;
  p.prior_theta = theta / p.Nm1 - ntheta / p.Nm1;
  p.prior_ntheta = ntheta / p.Nm1;
  p.theta_ratio = theta / ntheta - 1.0;
  p.log_ntheta = log(ntheta);
  p.log_small = log(0.01);
  return p;
}

/* r_prob / nor_x_theta of one gap: src/fast_painting.cpp:85-94 (and its two
 * repeats :125-134, :145-154). */
static void gap_terms(double rsum, const painter_t* p, double* rp, double* nx)
{
  *nx = -rsum + p->log_ntheta;
  *rp = 1.0 - exp(-rsum);
  if (*rp > 0.99) {
    *rp = 0.99;
    *nx = p->log_small + p->log_ntheta;
  }
}

/* Derived-site list of target k restricted to [first,last], both ends forced,
 * with the per-gap recombination terms: src/fast_painting.cpp:54-161 (whole
 * chunk) and :660-750 (one section).  rsum of a gap is accumulated left to
 * right starting from r[dk[j]]; the last entry is r[last] alone.
 * Returns nd; dk/rp/nx must hold last-first+2 entries. */
static int derived_sites(int N, const uint8_t* seq, const double* r, int k, int first, int last,
                         const painter_t* p, int* dk, double* rp, double* nx)
{
  int nd = 0;
  int s = first;
  for (;;) {
    dk[nd] = s;
    double rsum = r[s];
    if (s == last) {
      gap_terms(rsum, p, &rp[nd], &nx[nd]);
      nd++;
      break;
    }
    int q = s + 1;
    while (seq[(size_t)q * N + k] != '1' && q != last) {
      rsum += r[q];
      q++;
    }
    gap_terms(rsum, p, &rp[nd], &nx[nd]);
    nd++;
    s = q;
  }
  return nd;
}

/* Window bookkeeping: src/fast_painting.cpp:33-45, :73-82, :111-120, :163-166.
 * Walking consecutive derived sites (p,q): every window edge e = wb[w+1] with
 * p < e <= q closes window w at q and opens window w+1 at p. */
static void window_stones(int W, const int32_t* wb, int L, const int* dk, int nd, int* bb, int* be)
{
  int w = 0;
  bb[0] = 0;
  for (int j = 0; j + 1 < nd; j++) {
    const int pj = dk[j], qj = dk[j + 1];
    if (qj >= wb[w + 1] && pj < wb[w + 1]) {
      /* wb[W] == L > qj, so w never passes W-1 here */
      while (wb[w + 1] <= qj) {
        be[w] = qj;
        bb[w + 1] = pj;
        w++;
      }
    }
  }
  be[W - 1] = L - 1;
}

static void put_block(uint8_t* dst, int N, int32_t snp, float ls, const float* v)
{
  /* src/collapsed_matrix.hpp:240-252: u64 1, u64 N, i32 boundarySNP, f32 logscale, f32[N]. */
  uint64_t one = 1, n64 = (uint64_t)N;
  memcpy(dst, &one, 8);
  memcpy(dst + 8, &n64, 8);
  memcpy(dst + 16, &snp, 4);
  memcpy(dst + 20, &ls, 4);
  memcpy(dst + 24, v, (size_t)4 * N);
}

/* One target: src/fast_painting.cpp:18-632.  theta_p builds the painter's
 * constants, theta_d is Data::theta (used only in the backward pass, :438-440,
 * :488-489, :511-513): the two differ for the Rust caller (src/pipelines.rs:125,181).
 * rec(w) points at the (56+8N)-byte record of window w. */
static int paint_one(int N, int L, const uint8_t* seq, const double* r, double theta_p, double theta_d,
                     int W, const int32_t* wb, int k, uint8_t* out, size_t window_stride)
{
  const painter_t P = make_painter(N, theta_p);
  const double ntheta_d = 1.0 - theta_d;
  int* dk = (int*)malloc(sizeof(int) * (size_t)(L + 1));
  double* rp = (double*)malloc(sizeof(double) * (size_t)(L + 1));
  double* nx = (double*)malloc(sizeof(double) * (size_t)(L + 1));
  int* bb = (int*)malloc(sizeof(int) * (size_t)W);
  int* be = (int*)malloc(sizeof(int) * (size_t)W);
  double* a = (double*)malloc(sizeof(double) * (size_t)N);
  float* sa = (float*)malloc(sizeof(float) * (size_t)N * W);
  float* sb = (float*)malloc(sizeof(float) * (size_t)N * W);
  float* lsa = (float*)malloc(sizeof(float) * (size_t)W);
  float* lsb = (float*)malloc(sizeof(float) * (size_t)W);
  if (!dk || !rp || !nx || !bb || !be || !a || !sa || !sb || !lsa || !lsb) return 1;

  const int nd = derived_sites(N, seq, r, k, 0, L - 1, &P, dk, rp, nx);
  window_stones(W, wb, L, dk, nd, bb, be);
  const double rho_theta = 1.0 * P.theta_ratio + 1.0; /* derived*theta_ratio + 1.0, :305 */

  /* ---- forward, :214-408 ---- */
  {
    const uint8_t* row = seq;
    const uint8_t sk = row[k];
    for (int n = 0; n < N; n++) a[n] = (double)(sk > row[n]) * P.prior_theta + P.prior_ntheta;
    a[k] = 0.0;
    double S = 0.0;
    for (int n = 0; n < N; n++) S += a[n];
    double ls = 0.0;
    int w = 0;
    while (w < W && bb[w] == 0) {
      for (int n = 0; n < N; n++) sa[(size_t)w * N + n] = (float)a[n];
      lsa[w] = (float)ls;
      w++;
    }
    double rx = rp[0] / ((1.0 - rp[0]) * P.Nm1) * S;
    for (int j = 1; j < nd; j++) {
      const int s = dk[j];
      row = seq + (size_t)s * N;
      const uint8_t skj = row[k];
      ls += nx[j - 1];
      for (int n = 0; n < N; n++) {
        double v = a[n] + rx;
        v *= (skj > row[n]) ? rho_theta : 1.0;
        a[n] = v;
      }
      a[k] = 0.0;
      S = 0.0;
      for (int n = 0; n < N; n++) S += a[n];
      double rxn = S;
      if (rxn < P.lo || rxn > P.hi) {
        for (int n = 0; n < N; n++) a[n] /= S;
        ls += log(S);
        rxn = 1.0;
      }
      rxn *= rp[j] / ((1.0 - rp[j]) * P.Nm1);
      rx = rxn;
      while (w < W && bb[w] == s) {
        for (int n = 0; n < N; n++) sa[(size_t)w * N + n] = (float)a[n];
        lsa[w] = (float)ls;
        w++;
      }
    }
    if (w != W) return 2;
  }

  /* ---- backward, :410-599 ---- */
  {
    double ls = (double)log(P.Nm1) - nd * P.log_ntheta;
    const uint8_t* row = seq + (size_t)(L - 1) * N;
    uint8_t sk = row[k];
    for (int n = 0; n < N; n++) a[n] = 1.0;
    double B = 0.0;
    for (int n = 0; n < N; n++) B += (sk > row[n]) ? theta_d : ntheta_d;
    B -= ntheta_d;
    int w = W - 1;
    while (w >= 0 && be[w] == L - 1) {
      for (int n = 0; n < N; n++) sb[(size_t)w * N + n] = (float)a[n];
      lsb[w] = (float)ls;
      w--;
    }
    double rx = rp[nd - 1] / ((1.0 - rp[nd - 1]) * P.Nm1) * B;
    const uint8_t* row_next = row; /* row of the site processed before, sk belongs to it */
    for (int j = nd - 2; j >= 0; j--) {
      const int s = dk[j];
      ls += nx[j + 1];
      const double add_all = rx / ntheta_d;
      const double add_mis = rx / theta_d - add_all;
      for (int n = 0; n < N; n++) {
        const double d = (double)(sk > row_next[n]);
        double v = a[n] + d * add_mis + add_all;
        v *= d * P.theta_ratio + 1.0;
        a[n] = v;
      }
      row = seq + (size_t)s * N;
      sk = row[k];
      a[k] = 0.0;
      B = 0.0;
      for (int n = 0; n < N; n++) {
        if (sk > row[n]) B += theta_d * a[n];
        else B += ntheta_d * a[n];
      }
      double rxn = B;
      if (rxn < P.lo || rxn > P.hi) {
        for (int n = 0; n < N; n++) a[n] /= B;
        ls += oracle_fast_log((float)B); /* float approximation here, :562 */
        rxn = 1.0;
      }
      rxn *= rp[j] / ((1.0 - rp[j]) * P.Nm1);
      rx = rxn;
      while (w >= 0 && be[w] == s) {
        for (int n = 0; n < N; n++) sb[(size_t)w * N + n] = (float)a[n];
        lsb[w] = (float)ls;
        w--;
      }
      row_next = row;
    }
    if (w != -1) return 3;
  }

  /* ---- records, :603-615 ---- */
  for (int w = 0; w < W; w++) {
    uint8_t* rec = out + (size_t)w * window_stride;
    int32_t start = wb[w], end = wb[w + 1] - 1;
    memcpy(rec, &start, 4);
    memcpy(rec + 4, &end, 4);
    put_block(rec + 8, N, bb[w], lsa[w], sa + (size_t)w * N);
    put_block(rec + 8 + 24 + (size_t)4 * N, N, be[w], lsb[w], sb + (size_t)w * N);
  }
  free(dk); free(rp); free(nx); free(bb); free(be); free(a); free(sa); free(sb); free(lsa); free(lsb);
  return 0;
}

/* Batched, file-free painting of targets [k_begin,k_end): the loop of
 * pipeline/Paint.cpp:78-82.  seq is the L x N '0'/'1' matrix (row = SNP), r is
 * already multiplied by rho.  out is window-major: record (w,k) at
 * ((w*(k_end-k_begin)) + (k-k_begin)) * (56+8N). */
int oracle_paint_stones(int N, int L, const uint8_t* seq, const double* r, double theta_painter,
                        double theta_data, int W, const int32_t* wb, int k_begin, int k_end, uint8_t* out)
{
  if (N < 2 || L < 2 || W < 1 || wb[W] != L || k_begin < 0 || k_end > N || k_begin > k_end) return -1;
  const size_t R = 56 + (size_t)8 * N;
  const size_t K = (size_t)(k_end - k_begin);
  for (int k = k_begin; k < k_end; k++) {
    int rc = paint_one(N, L, seq, r, theta_painter, theta_data, W, wb, k,
                       out + (size_t)(k - k_begin) * R, K * R);
    if (rc) return rc;
  }
  return 0;
}

/* Number of derived sites of every target in [k_begin,k_end) (site 0 and L-1
 * forced): the count that sizes U_actual = 2 (N-1) sum nd_k. */
int oracle_count_derived(int N, int L, const uint8_t* seq, int k_begin, int k_end, int32_t* nd)
{
  for (int k = k_begin; k < k_end; k++) {
    int c = 2;
    for (int s = 1; s < L - 1; s++) c += seq[(size_t)s * N + k] == '1';
    nd[k - k_begin] = c;
  }
  return 0;
}

/* src/fast_painting.cpp:654-1126.  Posterior of target k inside one section
 * [first,last] seeded by the stones.  theta builds the painter constants AND is
 * Data::theta (the callers of this function construct the painter from the
 * same Data).  topology must hold (last-first+1)*N floats, logscales
 * last-first+1; returns the number of rows written, or <0. */
int oracle_repaint_section(int N, int L, const uint8_t* seq, const double* r, double theta, int first,
                           int last, const float* alpha_begin, const float* beta_end, float logscale_alpha,
                           float logscale_beta, int k, float* topology, float* logscales)
{
  (void)L;
  const painter_t P = make_painter(N, theta);
  const double ntheta = 1.0 - theta;
  const int cap = last - first + 2;
  int* dk = (int*)malloc(sizeof(int) * (size_t)cap);
  double* rp = (double*)malloc(sizeof(double) * (size_t)cap);
  double* nx = (double*)malloc(sizeof(double) * (size_t)cap);
  if (!dk || !rp || !nx) return -1;
  const int nd = derived_sites(N, seq, r, k, first, last, &P, dk, rp, nx);
  double* A = (double*)malloc(sizeof(double) * (size_t)nd * N);
  double* b = (double*)malloc(sizeof(double) * (size_t)N);
  if (!A || !b) return -1;
  const double rho_theta = 1.0 * P.theta_ratio + 1.0;

  /* forward, :786-919 */
  logscales[0] = logscale_alpha;
  for (int n = 0; n < N; n++) A[n] = (double)alpha_begin[n];
  A[k] = 0.0;
  double S = 0.0;
  for (int n = 0; n < N; n++) S += A[n];
  double rx = rp[0] / ((1.0 - rp[0]) * P.Nm1) * S;
  double prev = (double)logscales[0];
  for (int j = 1; j < nd; j++) {
    const uint8_t* row = seq + (size_t)dk[j] * N;
    const uint8_t sk = row[k];
    const double* ap = A + (size_t)(j - 1) * N;
    double* an = A + (size_t)j * N;
    prev += nx[j - 1];
    logscales[j] = (float)prev;
    for (int n = 0; n < N; n++) {
      double v = ap[n] + rx;
      v *= (sk > row[n]) ? rho_theta : 1.0;
      an[n] = v;
    }
    an[k] = 0.0;
    S = 0.0;
    for (int n = 0; n < N; n++) S += an[n];
    double rxn = S;
    if (rxn < P.lo || rxn > P.hi) {
      for (int n = 0; n < N; n++) an[n] /= S;
      prev += log(S);
      logscales[j] = (float)((double)logscales[j] + log(S));
      rxn = 1.0;
    }
    rxn *= rp[j] / ((1.0 - rp[j]) * P.Nm1);
    rx = rxn;
  }

  /* backward, :921-1107 */
  logscales[nd - 1] = logscales[nd - 1] + logscale_beta; /* float + float, :929 */
  const uint8_t* row = seq + (size_t)last * N;
  uint8_t sk = row[k];
  for (int n = 0; n < N; n++) b[n] = (double)beta_end[n];
  b[k] = 0.0;
  double B = 0.0;
  for (int n = 0; n < N; n++) B += ((sk > row[n]) ? theta : ntheta) * b[n];
  for (int n = 0; n < N; n++) topology[(size_t)(nd - 1) * N + n] = (float)(A[(size_t)(nd - 1) * N + n] * b[n]);
  rx = rp[nd - 1] / ((1.0 - rp[nd - 1]) * P.Nm1) * B;
  prev = (double)logscale_beta;
  const uint8_t* row_next = row;
  for (int j = nd - 2; j >= 0; j--) {
    prev += nx[j + 1];
    logscales[j] = (float)((double)logscales[j] + prev);
    const double add_all = rx / ntheta;
    const double add_mis = rx / theta - add_all;
    for (int n = 0; n < N; n++) {
      const double d = (double)(sk > row_next[n]);
      double v = b[n] + d * add_mis + add_all;
      v *= d * P.theta_ratio + 1.0;
      b[n] = v;
    }
    row = seq + (size_t)dk[j] * N;
    sk = row[k];
    b[k] = 0.0;
    B = 0.0;
    for (int n = 0; n < N; n++) {
      if (sk > row[n]) B += theta * b[n];
      else B += ntheta * b[n];
    }
    for (int n = 0; n < N; n++) topology[(size_t)j * N + n] = (float)(A[(size_t)j * N + n] * b[n]);
    double rxn = B;
    if (rxn < P.lo || rxn > P.hi) {
      for (int n = 0; n < N; n++) b[n] /= B;
      prev += log(B);
      logscales[j] = (float)((double)logscales[j] + log(B));
      rxn = 1.0;
    }
    rxn *= rp[j] / ((1.0 - rp[j]) * P.Nm1);
    rx = rxn;
    row_next = row;
  }
  free(dk); free(rp); free(nx); free(A); free(b);
  return nd;
}
