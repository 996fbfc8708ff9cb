// MLPG: maximum-likelihood parameter generation for the reference's three windows.
// Replaces MLPG.generation (idiaptts/misc/mlpg.py:94-127), i.e. the 62 python-level bandmat
// calls per utterance (build_poe :57-92, bla.solveh :125), by ONE launch over all
// (utterance, dimension) pairs of a batch.
//
// Math (per dimension, T frames): P x = b with the symmetric pentadiagonal precision matrix
//   P = diag(t0) + W1^T diag(t1) W1 + W2^T diag(t2) W2,   b = W0^T(m0 t0) + W1^T(m1 t1) + W2^T(m2 t2)
// W1 = [-.5 0 .5], W2 = [1 -2 1] Toeplitz, t_w = 1/var_w with the delta variances of the first
// and last frame forced to 1e11 (mlpg.py:114-117).  Solved by banded Cholesky (what
// bandmat.linalg.solveh does): forward sweep stores (d, l1, l2, y) per frame, backward sweep
// substitutes.  One lane owns one (utterance, dimension); a wave owns 64 neighbouring
// dimensions so every row access is one coalesced 512-B segment.
//
// Roofline: HBM.  Algorithmic bytes per frame = 187*8 read + 63*8 written = 2000 B
// (SURVEY.md section 8d); the factor scratch adds 3 f64 written + 4 read per (frame, dim).
#include <algorithm>

#include "common.h"

namespace itts {

constexpr double kBigVar = 100000000000.0;  // mlpg.py:114

struct MlpgArgs {
  const double* feat;
  int64_t ld_feat;
  int col0;
  int dim;
  const double* var;
  const int64_t* offsets;  // device copy, [U+1]
  double* out;
  int64_t ld_out;
  int ocol0;
  double* scratch;  // 3 planes [Ttot, dim]: 1/d, l1, l2 (shared factor) + nconv
  int64_t t_total;
  int* nconv;       // [dim] frame index where the shared factor becomes stationary
  // time-parallel solve (see mlpg_chunk_kernel)
  double* mf;       // [kmax][4][dim] forward transfer matrices of the shared-factor chunks
  double* mb;       // [kmax][4][dim] backward transfer matrices
  double* ends;     // [slots][2][dim] chunk end states of the sweep in progress
  int kmax;
};

constexpr int MLPG_CL = 64;   // frames per chunk of the time-parallel solve

// chunks of an utterance of T frames: all of length MLPG_CL except the last one, which takes the
// remainder plus a full chunk so that it always holds the two re-derived tail frames
__device__ __host__ inline int mlpg_num_chunks(int64_t T) {
  return (int)std::max<int64_t>(1, (T - 2) / MLPG_CL);
}

// The Cholesky factor of P depends on the variances and on the frame index only (not on the
// data), and -- because the delta variances are constant except in the first and last frame --
// it is the SAME for every utterance up to frame T-3.  mlpg_factor_kernel computes that shared
// factor once per dimension for the longest utterance ("T = infinity": edge variance at frame 0
// only); the per-utterance solve reads it and only re-derives the last two frames.  The solve is
// then two first-order-dependent sweeps of ~3 FMAs per frame instead of a sqrt and three
// divisions per frame in the dependency chain.
__global__ __launch_bounds__(64) void mlpg_factor_kernel(MlpgArgs a, int t_max) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  if (d >= a.dim) return;
  const int D = a.dim;
  const double v0 = a.var[d], v1 = a.var[D + d], v2 = a.var[2 * D + d];
  const double tau0 = 1.0 / v0, tau1_in = 1.0 / v1, tau2_in = 1.0 / v2, tau_edge = 1.0 / kBigVar;
  auto tau1 = [&](int t) -> double { return t < 0 ? 0.0 : (t == 0 ? tau_edge : tau1_in); };
  auto tau2 = [&](int t) -> double { return t < 0 ? 0.0 : (t == 0 ? tau_edge : tau2_in); };
  const int64_t plane = (int64_t)t_max * D;
  double* fd = a.scratch + d;
  double* fl1 = fd + plane;
  double* fl2 = fl1 + plane;
  double l1p = 0.0, l2p = 0.0, cprev = 0.0;
  int j = 0;
  for (; j < t_max; ++j) {
    const double pjj = tau0 + 0.25 * (tau1(j - 1) + tau1(j + 1)) + (tau2(j - 1) + 4.0 * tau2(j) + tau2(j + 1));
    const double pj1 = -2.0 * (tau2(j) + tau2(j + 1));
    const double pj2 = tau2(j + 1) - 0.25 * tau1(j + 1);
    const double dd = sqrt(pjj - l1p * l1p - l2p * l2p);
    const double l1 = (pj1 - cprev * l1p) / dd;
    const double l2 = pj2 / dd;
    fd[(int64_t)j * D] = 1.0 / dd;   // reciprocal: the solve multiplies instead of dividing
    fl1[(int64_t)j * D] = l1;
    fl2[(int64_t)j * D] = l2;
    // P is constant for j >= 2, so the recurrence is a fixed map of (l1p, l2p, cprev): once the
    // state repeats every later frame has the same factor -> stop (the solve clamps its factor
    // index to this frame)
    // (bit-for-bit repetition may never come: the rounded map can settle into a two-value cycle one
    // ulp wide, so "repeats to within 2^-50" ends the search; later frames reuse this factor)
    auto same = [](double x, double y) { return fabs(x - y) <= 8.9e-16 * fabs(y); };
    const bool fixed = j >= 3 && same(l1, l1p) && same(l2, cprev) && same(cprev, l2p);
    l2p = cprev;
    l1p = l1;
    cprev = l2;
    if (fixed) break;
  }
  a.nconv[d] = j < t_max ? j : t_max - 1;
}

// Latency-bound sequential sweeps: what limits throughput is the number of independent chains in
// flight, so a workgroup carries only LANES (16) dimensions -- a quarter-filled wave per
// workgroup, 128-B row segments -- which quadruples the waves (and outstanding loads) per batch.
constexpr int MLPG_LANES = 16;
__global__ __launch_bounds__(MLPG_LANES) void mlpg_kernel(MlpgArgs a, int t_max) {
  const int d = blockIdx.x * MLPG_LANES + threadIdx.x;
  const int u = blockIdx.y;
  if (d >= a.dim) return;
  const int64_t t0 = a.offsets[u];
  const int64_t T = a.offsets[u + 1] - t0;
  if (T <= 0) return;
  const int D = a.dim;
  const double v0 = a.var[d], v1 = a.var[D + d], v2 = a.var[2 * D + d];
  const double tau0 = 1.0 / v0;
  const double tau1_in = 1.0 / v1, tau2_in = 1.0 / v2, tau_edge = 1.0 / kBigVar;

  const double* f = a.feat + t0 * a.ld_feat + a.col0 + d;
  double* o = a.out + t0 * a.ld_out + a.ocol0 + d;
  const int64_t plane = (int64_t)t_max * D;
  const double* fd = a.scratch + d;
  const double* fl1 = fd + plane;
  const double* fl2 = fl1 + plane;

  auto tau1 = [&](int64_t t) -> double {
    if (t < 0 || t >= T) return 0.0;
    return (t == 0 || t == T - 1) ? tau_edge : tau1_in;
  };
  auto tau2 = [&](int64_t t) -> double {
    if (t < 0 || t >= T) return 0.0;
    return (t == 0 || t == T - 1) ? tau_edge : tau2_in;
  };

  const int64_t ncv = a.nconv[d];
  const double rv0 = 1.0 / v0, rv1 = 1.0 / v1, rv2 = 1.0 / v2, rvb = 1.0 / kBigVar;
  auto rvar1 = [&](int64_t t) { return (t == 0 || t == T - 1) ? rvb : rv1; };
  auto rvar2 = [&](int64_t t) { return (t == 0 || t == T - 1) ? rvb : rv2; };
  // factor of frame j: shared for j <= T-3, re-derived with the true edge variances for the last
  // two frames (tail[0] = frame T-2, tail[1] = frame T-1; T < 3: everything re-derived)
  double tl_d[2] = {1.0, 1.0}, tl_1[2] = {0.0, 0.0}, tl_2[2] = {0.0, 0.0};
  const int64_t n_shared = T >= 3 ? T - 2 : 0;

  // b-frames (mean / var, mlpg.py:123) of rows j-1, j, j+1 for windows 1 and 2.
  double p1 = 0.0, p2 = 0.0;  // row j-1
  double c0, c1, c2;          // row j
  c0 = f[0] * rv0;
  c1 = f[D] * rvar1(0);
  c2 = f[2 * D] * rvar2(0);
  // Cholesky state: row j entries L[j,j-1], L[j,j-2]; y[j-1], y[j-2]
  double l1p = 0.0, l2p = 0.0, cprev = 0.0, y1 = 0.0, y2 = 0.0;

  constexpr int PF = 8;  // rows prefetched ahead of the recurrence
  double nb0[PF], nb1[PF], nb2[PF], nd[PF], nl1[PF], nl2[PF];
  auto load_block = [&](int64_t jb, double (&b0)[PF], double (&b1)[PF], double (&b2)[PF],
                        double (&bd)[PF], double (&bl1)[PF], double (&bl2)[PF]) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int64_t t = jb + 1 + i;  // mean row j+1
      const int64_t tc = t < T ? t : T - 1;
      const double* r = f + tc * a.ld_feat;
      b0[i] = r[0];
      b1[i] = r[D];
      b2[i] = r[2 * D];
      const int64_t j = jb + i;      // factor of frame j
      const int64_t jc = j < n_shared ? (j < ncv ? j : ncv) : 0;
      bd[i] = fd[jc * D];
      bl1[i] = fl1[jc * D];
      bl2[i] = fl2[jc * D];
    }
  };
  load_block(0, nb0, nb1, nb2, nd, nl1, nl2);

  for (int64_t jb = 0; jb < T; jb += PF) {
    // issue the loads of the next block before touching the recurrence
    double fb0[PF], fb1[PF], fb2[PF], fbd[PF], fbl1[PF], fbl2[PF];
    load_block(jb + PF, fb0, fb1, fb2, fbd, fbl1, fbl2);
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int64_t j = jb + i;
      if (j < T) {
        double n0 = 0.0, n1 = 0.0, n2 = 0.0;
        if (j + 1 < T) {
          n0 = nb0[i] * rv0;
          n1 = nb1[i] * rvar1(j + 1);
          n2 = nb2[i] * rvar2(j + 1);
        }
        const double b = c0 + 0.5 * (p1 - n1) + (p2 - 2.0 * c2 + n2);
        double dd, l1, l2;  // dd holds 1 / L[j,j]
        if (j < n_shared) {
          dd = nd[i];
          l1 = nl1[i];
          l2 = nl2[i];
        } else {
          const double pjj = tau0 + 0.25 * (tau1(j - 1) + tau1(j + 1)) +
                             (tau2(j - 1) + 4.0 * tau2(j) + tau2(j + 1));
          const double pj1 = (j + 1 < T) ? -2.0 * (tau2(j) + tau2(j + 1)) : 0.0;
          const double pj2 = (j + 2 < T) ? (tau2(j + 1) - 0.25 * tau1(j + 1)) : 0.0;
          dd = 1.0 / sqrt(pjj - l1p * l1p - l2p * l2p);
          l1 = (pj1 - cprev * l1p) * dd;  // L[j+1,j]
          l2 = pj2 * dd;                  // L[j+2,j]
          const int q = (int)(j - (T - 2));  // 0 or 1 (or negative for T < 2: only frame T-1 => q = 1)
          if (q == 0) { tl_d[0] = dd; tl_1[0] = l1; tl_2[0] = l2; }
          else { tl_d[1] = dd; tl_1[1] = l1; tl_2[1] = l2; }
        }
        const double y = (b - l1p * y1 - l2p * y2) * dd;
        o[j * a.ld_out] = y;
        // advance to row j+1
        l2p = cprev;  // L[j+1,j-1]
        l1p = l1;
        cprev = l2;
        y2 = y1;
        y1 = y;
        p1 = c1;
        p2 = c2;
        c0 = n0;
        c1 = n1;
        c2 = n2;
      }
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      nb0[i] = fb0[i]; nb1[i] = fb1[i]; nb2[i] = fb2[i];
      nd[i] = fbd[i]; nl1[i] = fbl1[i]; nl2[i] = fbl2[i];
    }
  }

  // backward substitution L^T x = y
  double x1 = 0.0, x2 = 0.0;
  for (int64_t jb = T - 1; jb >= 0; jb -= PF) {
    double rd[PF], r1[PF], r2[PF], ry[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int64_t j = jb - i;
      if (j >= 0) {
        const int64_t jc = j < n_shared ? (j < ncv ? j : ncv) : 0;
        rd[i] = fd[jc * D];
        r1[i] = fl1[jc * D];
        r2[i] = fl2[jc * D];
        ry[i] = o[j * a.ld_out];
        if (j >= n_shared) {
          const int q = (j == T - 1) ? 1 : 0;
          rd[i] = tl_d[q];
          r1[i] = tl_1[q];
          r2[i] = tl_2[q];
        }
      } else {
        rd[i] = 1.0;
        r1[i] = r2[i] = ry[i] = 0.0;
      }
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int64_t j = jb - i;
      if (j >= 0) {
        const double x = (ry[i] - r1[i] * x1 - r2[i] * x2) * rd[i];
        o[j * a.ld_out] = x;
        x2 = x1;
        x1 = x;
      }
    }
  }
}

// ---- time-parallel solve -------------------------------------------------------------------------
// Both sweeps are second-order linear recurrences, y_j = (b_j - l1p_j y_{j-1} - l2p_j y_{j-2}) / d_j
// and its mirror image, so they parallelise over time as an affine prefix scan: cut every
// utterance into chunks of MLPG_CL frames; a chunk maps the two values that enter it to the two
// that leave it by  s_out = M s_in + e,  where M (2 x 2) only depends on the factor -- i.e. on
// (dimension, chunk index), the factor being shared by all utterances -- and e is what the
// chunk produces from a zero state.
//   pass A  every chunk runs the recurrence from zero and keeps e           (parallel)
//   pass B  every chunk folds the (M, e) of the chunks before it into its true entry state (at
//           most ~30 tiny steps) and runs the recurrence again, now writing    (parallel)
// Twice the arithmetic of the sequential sweep, ~T/64 times the parallelism: a batch of 256
// utterances x 62 dimensions keeps ~300 k lanes busy instead of 16 k.  The arithmetic inside a
// chunk is the sequential algorithm's; only the entry state of a chunk is rounded differently.
__global__ __launch_bounds__(64) void mlpg_transfer_kernel(MlpgArgs a, int t_max) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  const int k = blockIdx.y;
  if (d >= a.dim) return;
  const int D = a.dim;
  const int64_t plane = (int64_t)t_max * D;
  const double* fd = a.scratch + d;
  const double* fl1 = fd + plane;
  const double* fl2 = fl1 + plane;
  const int64_t ncv = a.nconv[d];
  auto F = [&](const double* pl, int64_t j) { return j < 0 ? 0.0 : pl[(j < ncv ? j : ncv) * D]; };
  const int64_t j0 = (int64_t)k * MLPG_CL, j1 = j0 + MLPG_CL;
  // forward: columns of M are the images of (y_{j0-1}, y_{j0-2}) = (1,0), (0,1)
  for (int c = 0; c < 2; ++c) {
    double y1 = c == 0 ? 1.0 : 0.0, y2 = c == 0 ? 0.0 : 1.0;
    for (int64_t j = j0; j < j1; ++j) {
      const double y = (-F(fl1, j - 1) * y1 - F(fl2, j - 2) * y2) * F(fd, j);
      y2 = y1;
      y1 = y;
    }
    a.mf[((int64_t)k * 4 + c) * D + d] = y1;        // M[0][c]
    a.mf[((int64_t)k * 4 + 2 + c) * D + d] = y2;    // M[1][c]
  }
  // backward: images of (x_{j1}, x_{j1+1}) on (x_{j0}, x_{j0+1})
  for (int c = 0; c < 2; ++c) {
    double x1 = c == 0 ? 1.0 : 0.0, x2 = c == 0 ? 0.0 : 1.0;
    for (int64_t j = j1 - 1; j >= j0; --j) {
      const double x = (-F(fl1, j) * x1 - F(fl2, j) * x2) * F(fd, j);
      x2 = x1;
      x1 = x;
    }
    a.mb[((int64_t)k * 4 + c) * D + d] = x1;
    a.mb[((int64_t)k * 4 + 2 + c) * D + d] = x2;
  }
}

// PASS 0: chunk end states from a zero entry state; PASS 1: true entry state, results written.
// BWD false: forward substitution (writes y into out); true: backward substitution (out: y -> x).
template <int PASS, bool BWD>
__global__ __launch_bounds__(64) void mlpg_chunk_kernel(MlpgArgs a, int t_max) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  const int k = blockIdx.y, u = blockIdx.z;
  if (d >= a.dim) return;
  const int64_t t0 = a.offsets[u];
  const int64_t T = a.offsets[u + 1] - t0;
  if (T <= 0) return;
  const int K = mlpg_num_chunks(T);
  if (k >= K) return;
  if (PASS == 0 && ((!BWD && k == K - 1) || (BWD && k == 0))) return;   // nobody reads that end state
  const int D = a.dim;
  const int64_t j0 = (int64_t)k * MLPG_CL, j1 = k == K - 1 ? T : j0 + MLPG_CL;
  const int64_t slot = t0 / MLPG_CL + u;                 // first end-state slot of this utterance
  double* E = a.ends + (slot * 2) * D + d;               // E[(c * 2 + i) * D]
  const int64_t plane = (int64_t)t_max * D;
  const double* fd = a.scratch + d;
  const double* fl1 = fd + plane;
  const double* fl2 = fl1 + plane;
  const int64_t ncv = a.nconv[d];
  const int64_t n_shared = T >= 3 ? T - 2 : 0;
  auto F = [&](const double* pl, int64_t j) { return j < 0 ? 0.0 : pl[(j < ncv ? j : ncv) * D]; };
  double* o = a.out + t0 * a.ld_out + a.ocol0 + d;

  // entry state: fold the chunks before this one (in sweep order)
  double s1 = 0.0, s2 = 0.0;
  if (PASS == 1) {
    if (!BWD) {
      for (int c = 0; c < k; ++c) {
        const double* M = a.mf + (int64_t)c * 4 * D + d;
        const double n1 = M[0] * s1 + M[D] * s2 + E[(c * 2) * D];
        const double n2 = M[2 * D] * s1 + M[3 * D] * s2 + E[(c * 2 + 1) * D];
        s1 = n1;
        s2 = n2;
      }
    } else {
      for (int c = K - 1; c > k; --c) {
        const double* M = a.mb + (int64_t)c * 4 * D + d;
        // the last chunk starts from zero, its M (which would need the tail factor) is not used
        const double m00 = c == K - 1 ? 0.0 : M[0], m01 = c == K - 1 ? 0.0 : M[D];
        const double m10 = c == K - 1 ? 0.0 : M[2 * D], m11 = c == K - 1 ? 0.0 : M[3 * D];
        const double n1 = m00 * s1 + m01 * s2 + E[(c * 2) * D];
        const double n2 = m10 * s1 + m11 * s2 + E[(c * 2 + 1) * D];
        s1 = n1;
        s2 = n2;
      }
    }
  }

  const double v0 = a.var[d], v1 = a.var[D + d], v2 = a.var[2 * D + d];
  const double tau0 = 1.0 / v0, tau1_in = 1.0 / v1, tau2_in = 1.0 / v2, tau_edge = 1.0 / kBigVar;
  auto tau1 = [&](int64_t t) -> double {
    if (t < 0 || t >= T) return 0.0;
    return (t == 0 || t == T - 1) ? tau_edge : tau1_in;
  };
  auto tau2 = [&](int64_t t) -> double {
    if (t < 0 || t >= T) return 0.0;
    return (t == 0 || t == T - 1) ? tau_edge : tau2_in;
  };
  // factor of frame j: shared table for j < n_shared, re-derived with the true edge variances for
  // the last two frames (always inside the last chunk)
  auto factor = [&](int64_t j, double l1p, double l2p, double cprev, double& dd, double& l1, double& l2) {
    if (j < n_shared) {
      dd = F(fd, j);
      l1 = F(fl1, j);
      l2 = F(fl2, j);
    } else {
      const double pjj = tau0 + 0.25 * (tau1(j - 1) + tau1(j + 1)) + (tau2(j - 1) + 4.0 * tau2(j) + tau2(j + 1));
      const double pj1 = (j + 1 < T) ? -2.0 * (tau2(j) + tau2(j + 1)) : 0.0;
      const double pj2 = (j + 2 < T) ? (tau2(j + 1) - 0.25 * tau1(j + 1)) : 0.0;
      dd = 1.0 / sqrt(pjj - l1p * l1p - l2p * l2p);
      l1 = (pj1 - cprev * l1p) * dd;
      l2 = pj2 * dd;
    }
  };

  if (!BWD) {
    const double* f = a.feat + t0 * a.ld_feat + a.col0 + d;
    const double rv0 = 1.0 / v0, rv1 = 1.0 / v1, rv2 = 1.0 / v2, rvb = 1.0 / kBigVar;
    auto rvar1 = [&](int64_t t) { return (t == 0 || t == T - 1) ? rvb : rv1; };
    auto rvar2 = [&](int64_t t) { return (t == 0 || t == T - 1) ? rvb : rv2; };
    // b-frame terms of rows j-1 (p), j (c), j+1 (n)
    double p1 = 0.0, p2 = 0.0;
    if (j0 > 0) {
      const double* r = f + (j0 - 1) * a.ld_feat;
      p1 = r[D] * rvar1(j0 - 1);
      p2 = r[2 * D] * rvar2(j0 - 1);
    }
    const double* rc = f + j0 * a.ld_feat;
    double c0 = rc[0] * rv0, c1 = rc[D] * rvar1(j0), c2 = rc[2 * D] * rvar2(j0);
    // Cholesky state entering row j0: L[j0,j0-1], L[j0,j0-2] and L[j0+1,j0-1]
    double l1p = F(fl1, j0 - 1), l2p = F(fl2, j0 - 2), cprev = F(fl2, j0 - 1);
    double y1 = s1, y2 = s2;
    // rows are prefetched PF at a time, the next block's loads issued before this block's chain
    constexpr int PF = 8;
    double nb0[PF], nb1[PF], nb2[PF], nd[PF], nl1[PF], nl2[PF];
    auto load_block = [&](int64_t jb, double (&b0)[PF], double (&b1)[PF], double (&b2)[PF],
                          double (&bd)[PF], double (&bl1)[PF], double (&bl2)[PF]) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int64_t t = jb + 1 + i;                      // data row j + 1
        const double* r = f + (t < T ? t : T - 1) * a.ld_feat;
        b0[i] = r[0];
        b1[i] = r[D];
        b2[i] = r[2 * D];
        const int64_t jf = jb + i;                         // factor of frame j
        const int64_t jc = jf < n_shared ? (jf < ncv ? jf : ncv) : 0;
        bd[i] = fd[jc * D];
        bl1[i] = fl1[jc * D];
        bl2[i] = fl2[jc * D];
      }
    };
    load_block(j0, nb0, nb1, nb2, nd, nl1, nl2);
    for (int64_t jb = j0; jb < j1; jb += PF) {
      double fb0[PF], fb1[PF], fb2[PF], fbd[PF], fbl1[PF], fbl2[PF];
      load_block(jb + PF, fb0, fb1, fb2, fbd, fbl1, fbl2);
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int64_t j = jb + i;
        if (j < j1) {
          double n0 = 0.0, n1 = 0.0, n2 = 0.0;
          if (j + 1 < T) {
            n0 = nb0[i] * rv0;
            n1 = nb1[i] * rvar1(j + 1);
            n2 = nb2[i] * rvar2(j + 1);
          }
          const double b = c0 + 0.5 * (p1 - n1) + (p2 - 2.0 * c2 + n2);
          double dd = nd[i], l1 = nl1[i], l2 = nl2[i];
          if (j >= n_shared) factor(j, l1p, l2p, cprev, dd, l1, l2);
          const double y = (b - l1p * y1 - l2p * y2) * dd;
          if (PASS == 1) o[j * a.ld_out] = y;
          l2p = cprev;
          l1p = l1;
          cprev = l2;
          y2 = y1;
          y1 = y;
          p1 = c1;
          p2 = c2;
          c0 = n0;
          c1 = n1;
          c2 = n2;
        }
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        nb0[i] = fb0[i]; nb1[i] = fb1[i]; nb2[i] = fb2[i];
        nd[i] = fbd[i]; nl1[i] = fbl1[i]; nl2[i] = fbl2[i];
      }
    }
    if (PASS == 0) {
      E[(k * 2) * D] = y1;
      E[(k * 2 + 1) * D] = y2;
    }
  } else {
    // the tail factors (frames T-2, T-1) need the Cholesky state that reaches them: two steps
    double tl_d[2] = {1.0, 1.0}, tl_1[2] = {0.0, 0.0}, tl_2[2] = {0.0, 0.0};
    if (k == K - 1) {
      double l1p = F(fl1, n_shared - 1), l2p = F(fl2, n_shared - 2), cprev = F(fl2, n_shared - 1);
      if (n_shared == 0) l1p = l2p = cprev = 0.0;
      for (int64_t j = n_shared; j < T; ++j) {
        double dd, l1, l2;
        factor(j, l1p, l2p, cprev, dd, l1, l2);
        const int q = (T - 1 - j) == 0 ? 1 : 0;     // frame T-1 -> slot 1, frame T-2 -> slot 0
        tl_d[q] = dd; tl_1[q] = l1; tl_2[q] = l2;
        l2p = cprev;
        l1p = l1;
        cprev = l2;
      }
    }
    double x1 = s1, x2 = s2;
    constexpr int PF = 8;
    for (int64_t jb = j1 - 1; jb >= j0; jb -= PF) {
      double rd[PF], r1[PF], r2[PF], ry[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int64_t j = jb - i;
        if (j >= j0) {
          const int64_t jc = j < n_shared ? (j < ncv ? j : ncv) : 0;
          rd[i] = fd[jc * D];
          r1[i] = fl1[jc * D];
          r2[i] = fl2[jc * D];
          ry[i] = o[j * a.ld_out];
          if (j >= n_shared) {
            const int q = (j == T - 1) ? 1 : 0;
            rd[i] = tl_d[q];
            r1[i] = tl_1[q];
            r2[i] = tl_2[q];
          }
        } else {
          rd[i] = 1.0;
          r1[i] = r2[i] = ry[i] = 0.0;
        }
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int64_t j = jb - i;
        if (j >= j0) {
          const double x = (ry[i] - r1[i] * x1 - r2[i] * x2) * rd[i];
          if (PASS == 1) o[j * a.ld_out] = x;
          x2 = x1;
          x1 = x;
        }
      }
    }
    if (PASS == 0) {
      E[(k * 2) * D] = x1;
      E[(k * 2 + 1) * D] = x2;
    }
  }
}

// np.gradient(x, axis=0) in float32 (misc/utils.py:103-105): one-sided at the ends, central
// inside; a single-frame utterance yields 0 (numpy raises there; the reference never hits it).
__global__ void gradient_f32_kernel(const float* x, int64_t ldx, float* out, int64_t ldo, int dim,
                                    const int64_t* offsets) {
  const int u = blockIdx.y;
  const int64_t t0 = offsets[u], T = offsets[u + 1] - t0;
  const int64_t n = T * dim;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / dim;
    const int d = (int)(i - t * dim);
    const float* c = x + (t0 + t) * ldx + d;
    float g;
    if (T == 1) {
      g = 0.f;
    } else if (t == 0) {
      g = c[ldx] - c[0];
    } else if (t == T - 1) {
      g = c[0] - c[-ldx];
    } else {
      g = (c[ldx] - c[-ldx]) / 2.0f;
    }
    out[(t0 + t) * ldo + d] = g;
  }
}

}  // namespace itts

using namespace itts;

extern "C" int64_t itts_mlpg_scratch_bytes(int64_t t_total, int dim) {
  if (t_total < 0 || dim <= 0) return 0;
  // 3 factor planes + device copy of the offsets (<= t_total + 1 entries, padded) + nconv, then the
  // time-parallel solve's transfer matrices (2 x kmax x 4 x dim) and chunk end states
  // (slots x 2 x dim with slots <= t_total / CL + n_utts + 1 <= t_total / CL + t_total + 2)
  const int64_t kmax = t_total / itts::MLPG_CL + 2;
  const int64_t slots = t_total / itts::MLPG_CL + t_total + 3;
  return 3 * t_total * (int64_t)dim * 8 + (t_total + 2) * 8 + ((int64_t)dim * 4 + 16) / 8 * 8 + 8 +
         (2 * kmax * 4 + slots * 2) * (int64_t)dim * 8;
}

extern "C" int itts_mlpg_generation(const double* d_feat, int64_t ld_feat, int col0, int dim,
                                    const double* d_var, const int64_t* h_offsets, int n_utts,
                                    double* d_out, int64_t ld_out, int ocol0, void* d_scratch,
                                    void* stream) {
  ITTS_REQUIRE(d_var && h_offsets && (n_utts == 0 || (d_feat && d_out && d_scratch)), "null pointer");
  ITTS_REQUIRE(dim > 0 && n_utts >= 0 && col0 >= 0 && ocol0 >= 0, "bad sizes");
  ITTS_REQUIRE(ld_feat >= col0 + 3 * (int64_t)dim && ld_out >= ocol0 + (int64_t)dim,
               "leading dimension too small");
  if (n_utts == 0) return ITTS_OK;
  const int64_t t_total = h_offsets[n_utts];
  ITTS_REQUIRE(h_offsets[0] == 0 && t_total >= 0, "offsets must start at 0");
  for (int u = 0; u < n_utts; ++u)
    ITTS_REQUIRE(h_offsets[u + 1] >= h_offsets[u], "offsets must be non-decreasing");
  ITTS_REQUIRE(n_utts <= t_total + 1, "more utterances than frames");
  if (t_total == 0) return ITTS_OK;
  hipStream_t s = as_stream(stream);
  double* scratch = reinterpret_cast<double*>(d_scratch);
  int64_t* d_off = reinterpret_cast<int64_t*>(scratch + 3 * t_total * (int64_t)dim);
  ITTS_HIP_CHECK(hipMemcpyAsync(d_off, h_offsets, (n_utts + 1) * sizeof(int64_t),
                                hipMemcpyHostToDevice, s));
  int* d_nconv = reinterpret_cast<int*>(d_off + (t_total + 2));
  MlpgArgs a{d_feat, ld_feat, col0, dim, d_var, d_off, d_out, ld_out, ocol0, scratch, t_total, d_nconv};
  int64_t t_max = 0;
  for (int u = 0; u < n_utts; ++u) t_max = std::max(t_max, h_offsets[u + 1] - h_offsets[u]);
  hipLaunchKernelGGL(mlpg_factor_kernel, dim3((dim + 63) / 64), dim3(64), 0, s, a, (int)t_max);
  ITTS_LAUNCH_CHECK();
  const int kchunks = mlpg_num_chunks(t_max);
  if (kchunks < 3) {      // short utterances: the sequential sweeps are as fast
    dim3 grid((dim + MLPG_LANES - 1) / MLPG_LANES, n_utts);
    hipLaunchKernelGGL(mlpg_kernel, grid, dim3(MLPG_LANES), 0, s, a, (int)t_max);
    ITTS_LAUNCH_CHECK();
    return ITTS_OK;
  }
  double* extra = reinterpret_cast<double*>(reinterpret_cast<char*>(d_nconv) + ((int64_t)dim * 4 + 16) / 8 * 8 + 8);
  a.kmax = (int)(t_total / MLPG_CL + 2);
  a.mf = extra;
  a.mb = a.mf + (int64_t)a.kmax * 4 * dim;
  a.ends = a.mb + (int64_t)a.kmax * 4 * dim;
  const dim3 tg((dim + 63) / 64, kchunks);
  hipLaunchKernelGGL(mlpg_transfer_kernel, tg, dim3(64), 0, s, a, (int)t_max);
  const dim3 cg((dim + 63) / 64, kchunks, n_utts);
  hipLaunchKernelGGL((mlpg_chunk_kernel<0, false>), cg, dim3(64), 0, s, a, (int)t_max);
  hipLaunchKernelGGL((mlpg_chunk_kernel<1, false>), cg, dim3(64), 0, s, a, (int)t_max);
  hipLaunchKernelGGL((mlpg_chunk_kernel<0, true>), cg, dim3(64), 0, s, a, (int)t_max);
  hipLaunchKernelGGL((mlpg_chunk_kernel<1, true>), cg, dim3(64), 0, s, a, (int)t_max);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_gradient_f32(const float* d_x, int64_t ld_x, float* d_out, int64_t ld_out,
                                 int dim, const int64_t* h_offsets, int n_utts, void* stream) {
  ITTS_REQUIRE(h_offsets && (n_utts == 0 || (d_x && d_out)), "null pointer");
  ITTS_REQUIRE(dim > 0 && ld_x >= dim && ld_out >= dim && n_utts >= 0, "bad sizes");
  if (n_utts == 0 || h_offsets[n_utts] == 0) return ITTS_OK;
  hipStream_t s = as_stream(stream);
  int64_t* d_off = nullptr;
  ITTS_HIP_CHECK(hipMallocAsync((void**)&d_off, (n_utts + 1) * sizeof(int64_t), s));
  ITTS_HIP_CHECK(hipMemcpyAsync(d_off, h_offsets, (n_utts + 1) * sizeof(int64_t),
                                hipMemcpyHostToDevice, s));
  int64_t maxT = 0;
  for (int u = 0; u < n_utts; ++u) maxT = std::max(maxT, h_offsets[u + 1] - h_offsets[u]);
  int bx = (int)std::min<int64_t>((maxT * dim + 255) / 256, 64);
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(gradient_f32_kernel, dim3(bx, n_utts), dim3(256), 0, s, d_x, ld_x, d_out,
                     ld_out, dim, d_off);
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(hipFreeAsync(d_off, s));
  return ITTS_OK;
}
