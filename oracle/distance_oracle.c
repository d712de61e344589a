/*
 * distance_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the distance matrix of tcztzy/relate's tree builder,
 * DistanceMeasure::GetMatrix (src/anc_builder.cpp:113-211), the part after the
 * topology of the section has been loaded (:120-196).  Only tests/ may load it.
 *
 * Parity status: PARITY UNPINNED for the matrix as a whole.  The reference's
 * anc_builder.cpp cannot be compiled in this container (anc_builder.hpp ->
 * mutations.hpp -> <gzstream.h>, a library the image does not have), and the
 * reference holds no fixture of a distance matrix; the only pinned ingredient is
 * fast_log (test/test_log.cpp:6-15, checked in tests/test_oracle.py).  This file
 * follows the reference's expression order and types line by line: weights in
 * double, the mix of the two topology rows in double, narrowed to float for
 * fast_log, float from there on, scale = -1.0f (src/anc_builder.hpp:46).
 *
 * Build: gcc -O2 -ffp-contract=off, linked into libpaint_oracle.so.
 */
#include <math.h>
#include <stdint.h>

float oracle_fast_log(float val);

/* top: rows x pitch floats, ls: rows floats.  For row n of the matrix: row[n] is the
 * topology row at or before the SNP (v_snp_prev), interp[n] says whether the reference
 * takes it as is (:122-141) or mixes it with the next one (:143-193). */
void oracle_distance_matrix(int N, const float* top, int pitch, const float* ls, const int64_t* row,
                            const uint8_t* interp, const double* wl, const double* wr, float* matrix)
{
  const float scale = -1.0f;
  for (int n = 0; n < N; n++) {
    float min = INFINITY;
    float* out = matrix + (int64_t)n * N;
    const float* tp = top + row[n] * pitch;
    if (!interp[n]) {
      const float logscale_prev = ls[row[n]];
      for (int m = 0; m < N; m++) {
        out[m] = (oracle_fast_log(tp[m]) + logscale_prev) * scale;
        if (out[m] < min) min = out[m];
      }
      out[n] = 0.0f;
    } else {
      const float* tn = tp + pitch;
      const double weight_left = wl[n], weight_right = wr[n];
      const float logscale_prev = ls[row[n]], logscale_next = ls[row[n] + 1];
      /* `exp(float)` written unqualified (:171-172): the C function on a double, narrowed on assignment */
      const float e_pn = (float)exp((double)(logscale_prev - logscale_next));
      const float e_np = (float)exp((double)(logscale_next - logscale_prev));
      for (int m = 0; m < N; m++) {
        if (logscale_prev <= logscale_next) {
          out[m] = (oracle_fast_log((float)(weight_left * tp[m] * e_pn + weight_right * tn[m])) + logscale_next) * scale;
        } else {
          out[m] = (oracle_fast_log((float)(weight_left * tp[m] + weight_right * tn[m] * e_np)) + logscale_prev) * scale;
        }
        if (out[m] < min) min = out[m];
      }
      out[n] = 0.0f;
    }
    for (int j = 0; j < N; j++)
      if (j != n) out[j] -= min;
  }
}
