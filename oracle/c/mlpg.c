/* ORACLE (test infrastructure; never linked into or called by the product path).
 *
 * CPU restatement of MLPG.generation, idiaptts/misc/mlpg.py:94-127 (reference), whose
 * arithmetic lives in the third-party Cython package `bandmat` (unpinned, requirements.txt:10;
 * not present in /root/reference): build_win_mats (mlpg.py:31-55) builds Toeplitz window
 * matrices, build_poe (mlpg.py:57-92) accumulates b = sum_w W_w^T (mu_w / var_w) and
 * P = sum_w W_w^T diag(1/var_w) W_w, bla.solveh (mlpg.py:125) is a banded Cholesky solve.
 * Pinned by the reference's benchmark known answer (test_AcousticModelTrainer.py:104) through
 * tests/test_oracle_golden.py and against scipy.linalg.solveh_banded.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* features [T, ldf] (static | delta | delta-delta blocks of width D starting at col0),
 * var [3D] diagonal of the covariance, out [T, ldo] at ocol0. */
int orc_mlpg(const double* feat, long T, long ldf, int col0, int D, const double* var, double* out,
             long ldo, int ocol0) {
  if (T <= 0) return 0;
  double* P0 = (double*)malloc(sizeof(double) * T * 6);
  if (!P0) return -1;
  double *P1 = P0 + T, *P2 = P1 + T, *b = P2 + T, *tau1 = b + T, *tau2 = tau1 + T;
  for (int d = 0; d < D; ++d) {
    const double v0 = var[d], v1 = var[D + d], v2 = var[2 * D + d];
    /* mlpg.py:111-117: delta variances of the first and last frame are 1e11 */
    for (long t = 0; t < T; ++t) {
      const int edge = (t == 0 || t == T - 1);
      tau1[t] = 1.0 / (edge ? 100000000000.0 : v1);
      tau2[t] = 1.0 / (edge ? 100000000000.0 : v2);
    }
    memset(P0, 0, sizeof(double) * T * 4);
    /* window 0: [1] */
    for (long t = 0; t < T; ++t) {
      b[t] += feat[t * ldf + col0 + d] / v0;
      P0[t] += 1.0 / v0;
    }
    /* window 1: row t of W1 = -0.5 at t-1, +0.5 at t+1 */
    for (long t = 0; t < T; ++t) {
      const int edge = (t == 0 || t == T - 1);
      const double bf = feat[t * ldf + col0 + D + d] / (edge ? 100000000000.0 : v1);
      const double ta = tau1[t];
      if (t - 1 >= 0) { b[t - 1] += -0.5 * bf; P0[t - 1] += 0.25 * ta; }
      if (t + 1 < T) { b[t + 1] += 0.5 * bf; P0[t + 1] += 0.25 * ta; }
      if (t - 1 >= 0 && t + 1 < T) P2[t - 1] += -0.25 * ta;
    }
    /* window 2: row t of W2 = 1 at t-1, -2 at t, 1 at t+1 */
    for (long t = 0; t < T; ++t) {
      const int edge = (t == 0 || t == T - 1);
      const double bf = feat[t * ldf + col0 + 2 * D + d] / (edge ? 100000000000.0 : v2);
      const double ta = tau2[t];
      if (t - 1 >= 0) { b[t - 1] += bf; P0[t - 1] += ta; }
      b[t] += -2.0 * bf; P0[t] += 4.0 * ta;
      if (t + 1 < T) { b[t + 1] += bf; P0[t + 1] += ta; }
      if (t - 1 >= 0) P1[t - 1] += -2.0 * ta;
      if (t + 1 < T) P1[t] += -2.0 * ta;
      if (t - 1 >= 0 && t + 1 < T) P2[t - 1] += ta;
    }
    /* banded Cholesky P = L L^T, L lower with two sub-diagonals stored in place:
     * P0[j] = L[j,j], P1[j] = L[j+1,j], P2[j] = L[j+2,j] */
    for (long j = 0; j < T; ++j) {
      double s = P0[j];
      if (j >= 1) s -= P1[j - 1] * P1[j - 1];
      if (j >= 2) s -= P2[j - 2] * P2[j - 2];
      if (s <= 0.0) { free(P0); return -2; }
      const double l = sqrt(s);
      P0[j] = l;
      if (j + 1 < T) {
        double a = P1[j];
        if (j >= 1) a -= P2[j - 1] * P1[j - 1];
        P1[j] = a / l;
      }
      if (j + 2 < T) P2[j] = P2[j] / l;
    }
    /* forward L y = b */
    for (long j = 0; j < T; ++j) {
      double s = b[j];
      if (j >= 1) s -= P1[j - 1] * b[j - 1];
      if (j >= 2) s -= P2[j - 2] * b[j - 2];
      b[j] = s / P0[j];
    }
    /* backward L^T x = y */
    for (long j = T - 1; j >= 0; --j) {
      double s = b[j];
      if (j + 1 < T) s -= P1[j] * b[j + 1];
      if (j + 2 < T) s -= P2[j] * b[j + 2];
      b[j] = s / P0[j];
      out[j * ldo + ocol0 + d] = b[j];
    }
  }
  free(P0);
  return 0;
}
