// MLPG: maximum-likelihood parameter generation for the reference's three windows.
// Replaces MLPG.generation (idiaptts/misc/mlpg.py:94-127), i.e. the 62 python-level bandmat
// calls per utterance (build_poe :57-92, bla.solveh :125), by one call over all
// (utterance, dimension) pairs of a batch.
//
// Math (per dimension, T frames): P x = b with the symmetric pentadiagonal precision matrix
//   P = diag(t0) + W1^T diag(t1) W1 + W2^T diag(t2) W2,   b = W0^T(m0 t0) + W1^T(m1 t1) + W2^T(m2 t2)
// W1 = [-.5 0 .5], W2 = [1 -2 1] Toeplitz, t_w = 1/var_w with the delta variances of the first
// and last frame forced to 1e11 (mlpg.py:114-117).  Solved by banded Cholesky (what
// bandmat.linalg.solveh does): forward sweep y = L^-1 b, backward sweep x = L^-T y.  One lane owns
// one (utterance, dimension); a wave owns 64 neighbouring dimensions so every row access is one
// coalesced 512-B segment.
//
// What is in this file:
//   mlpg_factor_kernel          the data-independent Cholesky factor, once per dimension, shared by
//                               all utterances (stops when it repeats)
//   mlpg_kernel                 the two sweeps frame by frame: batches of short utterances
//   mlpg_prep / reduce / scan / solve   everything else, no wait anywhere: the backward contribution of
//                               a chunk is accumulated while walking forward (adjoint identity), a
//                               two-level scan gives every chunk its entry states, a second pass solves
// (Rounds 2 and 3 also carried a four-pass chunked solve and a single-pass kernel with cross-workgroup
// waits; both measured slower -- DESIGN.md section 11c -- and left the library in round 4.)
//
// Roofline: HBM.  Algorithmic bytes per frame = 187*8 read + 63*8 written = 2000 B
// (SURVEY.md section 8d); measured traffic and rates: DESIGN.md section 11c.
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "context.h"

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
};

constexpr int MLPG_SEQ_BELOW = 194;   // utterances shorter than this take the sequential sweeps (three 64-frame chunks + tail)
constexpr int MLPG_RING_FROM = 128;     // (utterance, 64-dimension block) units from which the one-pass kernel takes over

// The Cholesky factor of P depends on the variances and on the frame index only (not on the
// data), and -- because the delta variances are constant except in the first and last frame --
// it is the SAME for every utterance up to frame T-3.  mlpg_factor_kernel computes that shared
// factor once per dimension for the longest utterance ("T = infinity": edge variance at frame 0
// only); the per-utterance solve reads it and only re-derives the last two frames.  The solve is
// then two first-order-dependent sweeps of ~3 FMAs per frame instead of a sqrt and three
// divisions per frame in the dependency chain.
// 1 / sqrt(x) for the pivot of the factor: hardware estimate + three Newton steps (nine dependent
// multiply-adds) instead of a square root and a division (~60 dependent instructions) -- the factor is one
// latency chain per dimension in front of every solve, 18-22 us of a 256-utterance call.  The estimate
// carries >= 13 bits, three steps square that past the 53 of a double; the last step's residual form keeps
// the result within an ulp or two of the correctly rounded one (the solve's 1e-10 budget against the
// oracle is nine orders above that).
__device__ __forceinline__ double factor_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  const double r = 0.5 - h * y * y;      // residual of the third step
  return y + y * r;
}

__device__ __forceinline__ void mlpg_factor_block(const MlpgArgs& a, int t_max, int block) {
  const int d = block * 64 + threadIdx.x;
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
    // one square root and one division per frame: this loop is a pure latency chain (one wave per
    // 64 dimensions) in front of every solve
    const double inv = factor_rsqrt(pjj - l1p * l1p - l2p * l2p);
    const double l1 = (pj1 - cprev * l1p) * inv;
    const double l2 = pj2 * inv;
    fd[(int64_t)j * D] = inv;        // reciprocal: the solve multiplies instead of dividing
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

__global__ __launch_bounds__(64) void mlpg_factor_kernel(MlpgArgs a, int t_max) {
  mlpg_factor_block(a, t_max, blockIdx.x);
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

// ---- one pass: the right-hand side never leaves the CU (round 5) ----------------------------------------------------
// A workgroup owns (utterance, 64 dimensions: all of them for the usual 60 + 1 + 1 streams).  Wave 0 walks the two
// sweeps frame by frame, a lane a dimension; the b / y / x rows it works on live in LDS -- a ring of RING_CAP frames x
// 64 doubles (147 KB) -- so that the recurrences' operands are LDS reads issued ahead of the chain (what mlpg_kernel
// pays per frame in memory latency is gone) and the input rows are read ONCE.  Seven helper waves work 24-frame
// segments around the sweep -- forward: the input rows of a segment a ring's length ahead -> b into the ring, after
// moving the y that occupied those slots (RING_CAP frames back) out to the output rows; backward: the finished
// segment's x out to the output rows, then the y of a ring's length further down back into the slots.  Progress words in
// LDS instead of barriers, no hand-off between workgroups, no scratch, ONE launch: the sweep derives the factor's
// moving head (the first 30 - 50 rows, until it repeats) itself while it walks them -- the same arithmetic as
// mlpg_factor_kernel, row by row -- and leaves the rows in the factor table for the way back.
//
// What shaped it (scripts/lat_lab, profiles/r5_mlpg_ring.md):
//  * a lone wave issues an instruction every 5 - 7 cycles whatever its width, so the sweep costs the same for 16
//    lanes as for 64: the first form of this kernel (16 dimensions x 1 152 frames a workgroup, the whole y of most
//    utterances in LDS) spent 4 x the sweep time of this one to save the y round trip.  Here y makes the trip (out and
//    back through the output rows, last in first out): 48 bytes per frame and dimension where the algorithm needs
//    32 and the three launch form moves 65;
//  * what a frame costs the sweep is its instruction COUNT: a segment that lies on the stationary factor altogether
//    is straight-line code, 4 instructions a frame; everything else (the head, the segment with the tail frames)
//    runs in ROLLED loops -- unrolled they were 40 KB of code that runs once a workgroup, every line of it an
//    instruction-cache miss behind the helpers' streams (40 us for the first segment);
//  * the CU has ONE memory pipeline: a load the sweep waits for queues behind whatever the seven helpers have asked
//    for (5 us on the way forward).  The sweep therefore does not wait for loads: the head's factor rows are derived on
//    the way forward; on the way back -- the helpers only store by then, a trip is the L2's 0.5 - 1 us -- they come
//    from the table twelve rows ahead of their use;
//  * every row of the factor's head is a frame off the straight-line path: what a solve takes followed the VARIANCES
//    (21 to 270 rows until the factor repeats) until the head's own cost was cut (last paragraph);
//  * the helpers' segment, row and edge arithmetic belongs on the scalar unit (wave number through readfirstlane).
// Arithmetic: mlpg_kernel's, expression for expression.
#ifndef MLPG_RING_NT
#define MLPG_RING_NT 0
#endif
// cache policy of the streams (bits: 1 input rows, 2 x stores, 4 y read back, 8 y parked): non-temporal where set.
// Measured (scripts/r5_job27.sh, every variant on one box, same variances): none of them matters at 4 096 utterances or
// with float32 rows; float64 rows at 256 utterances 258 -> 247 us with the INPUT rows non-temporal -- the y that is
// out (156 MB) then survives in the 256-MB memory-side cache until it comes back.  The kernel's NT_IN takes that.
#define RING_LD(bit, p) (((MLPG_RING_NT & (bit)) || ((bit) == 1 && NT_IN)) ? __builtin_nontemporal_load(p) : *(p))
#define RING_ST(bit, p, v) do { if (MLPG_RING_NT & (bit)) __builtin_nontemporal_store((v), (p)); else *(p) = (v); } while (0)
constexpr int RING_LANES = 64, RING_SEG = 24, RING_CAP = 288, RING_HELPERS = 7, RING_THREADS = 64 * (1 + RING_HELPERS);
constexpr int RING_LDS_BYTES = RING_CAP * RING_LANES * 8 + 128;     // + progress words
static_assert(RING_CAP % RING_SEG == 0 && RING_SEG % 8 == 0, "ring geometry");
struct RingArgs {
  MlpgArgs a;
  const int64_t* bounds;  // [workgroup rank][2]: first frame, end frame of its utterance (utterances longest first)
  int t_max;
  const float* feat32;    // the input rows when they are float32 (itts_mlpg_generation_f32): a.feat is unused then
};
// progress words in LDS (one writer each; release / acquire at workgroup scope)
__device__ __forceinline__ void ring_post(int* w, int v) { __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int ring_peek(const int* w) { return __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// FT: the type of the input rows (double, or float: the network's own output type -- converted in the load, which is exact)
// WIDE: the helpers move two dimensions a lane and two rows an instruction (an even number of dimensions)
// NT_IN: the input rows are read with the non-temporal hint (see RING_LD)
template <typename FT, bool WIDE, bool NT_IN>
__global__ __launch_bounds__(RING_THREADS) void mlpg_ring_kernel(RingArgs g) {
  extern __shared__ __attribute__((aligned(16))) char rsm[];
  double* ring = reinterpret_cast<double*>(rsm);          // [RING_CAP][64]
  int* prog = reinterpret_cast<int*>(rsm + RING_CAP * RING_LANES * 8);
  // prog[0]: forward sweep: segments finished            prog[1 + h]: helper h, forward: its segments prepared (count)
  // prog[8]: the backward sweep has begun                prog[9 + h]: helper h, backward: its segments stored / refilled (count)
  // prog[16]: backward sweep: lowest segment finished (n_segments: none yet)
  const MlpgArgs& a = g.a;
  const int blk = blockIdx.x;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // the utterances' bounds are read in place from page-locked host memory (no copy on the stream in front of the
  // launch): one thread fetches this workgroup's pair -- one 16-byte load, one trip over the bus --, the others get it
  // through LDS
  int64_t* bounds = reinterpret_cast<int64_t*>(prog + 24);
  if (tid == 0) {
    typedef int64_t Pair __attribute__((ext_vector_type(2)));
    const Pair b = reinterpret_cast<const Pair*>(g.bounds)[blockIdx.y];
    bounds[0] = b.x;
    bounds[1] = b.y;
  }
  if (tid < 24) prog[tid] = 0;
  __syncthreads();
  const int64_t t0 = bounds[0];
  const int T = (int)(bounds[1] - t0);            // (frames of one utterance: 32 bits)
  if (T <= 0) return;
  const int D = a.dim;
  const double rvb = 1.0 / kBigVar;
  const int nseg = (T + RING_SEG - 1) / RING_SEG;
  constexpr int ring_segs = RING_CAP / RING_SEG;
  // slot of the first frame of segment sgm (the ring holds a whole number of segments: the frames of a segment sit in
  // consecutive slots)
  auto seg_slot = [](int sgm) { return (sgm % ring_segs) * RING_SEG; };
  const int dc = blk * RING_LANES + lane < D ? blk * RING_LANES + lane : D - 1;
  const int64_t plane = (int64_t)g.t_max * D;                 // the factor table: three planes of t_max rows

  if (wave != 0) {
    // ================= helper wave h: segments q = h, h + H, h + 2 H ..  (a lane: one dimension)
    // (the wave number through readfirstlane: segment numbers, row numbers and the edge tests are then the scalar
    // unit's work and the branches on them branches of the wave -- left as threadIdx arithmetic they were vector
    // selects around every load and every frame: 600 cycles a frame, 15 us a segment)
    const int h = wave - 1;
    const int hd = blk * RING_LANES + lane;
    const double hrv0 = 1.0 / a.var[dc], hrv1 = 1.0 / a.var[D + dc], hrv2 = 1.0 / a.var[2 * D + dc];
    const FT* hf = (std::is_same<FT, float>::value ? reinterpret_cast<const FT*>(g.feat32) : reinterpret_cast<const FT*>(a.feat)) +
                   t0 * a.ld_feat + a.col0 + dc;
    double* o = a.out + t0 * a.ld_out + a.ocol0 + dc;
    auto r1 = [&](int t) { return (t == 0 || t == T - 1) ? rvb : hrv1; };
    auto r2 = [&](int t) { return (t == 0 || t == T - 1) ? rvb : hrv2; };
    // ---- forward: input rows -> b into the ring, never more than a ring's length ahead of the sweep
    int mine = 0;
    // one segment; INNER: no frame of it is, or neighbours, an edge of the utterance (no row clamps, no edge variances)
    auto forward_segment = [&](int q, auto inner_tag) {
      constexpr bool INNER = decltype(inner_tag)::value;
      const int j0 = q * RING_SEG;
      double* base = ring + seg_slot(q) * RING_LANES + lane;
      // the whole segment's loads in flight together, and BEFORE the wait for its slots (registers are the only place
      // they need; seven helpers x 24 frames under way whatever the sweep is doing): the static column of rows j0 ..
      // j0 + 23, the delta and delta-delta columns of rows j0 - 1 .. j0 + 24 (each row serves as a frame's own and as
      // both its neighbours'), row numbers held inside the utterance
      double st[RING_SEG], d1[RING_SEG + 2], d2[RING_SEG + 2];
#pragma unroll
      for (int i = 0; i < RING_SEG + 2; ++i) {
        int r = j0 - 1 + i;
        if (!INNER) r = r < 0 ? 0 : (r < T ? r : T - 1);
        const FT* row = hf + (int64_t)r * a.ld_feat;
        d1[i] = (double)RING_LD(1, row + D);
        d2[i] = (double)RING_LD(1, row + 2 * D);
        if (i >= 1 && i <= RING_SEG) st[i - 1] = (double)RING_LD(1, row);
      }
      while (q - ring_peek(prog) >= ring_segs) __builtin_amdgcn_s_sleep(2);      // the sweep has left segment q - ring_segs
      if (j0 >= RING_CAP && hd < D) {                   // the y of frames j0 - RING_CAP .. leave the ring
#pragma unroll
        for (int i = 0; i < RING_SEG; ++i)
          if (INNER || j0 + i < T) RING_ST(8, o + (int64_t)(j0 + i - RING_CAP) * a.ld_out, base[i * RING_LANES]);
      }
#pragma unroll
      for (int i = 0; i < RING_SEG; ++i) {
        const int j = j0 + i;
        if (INNER || j < T) {
          double bj;
          if (INNER) {
            const double c0 = st[i] * hrv0, c2 = d2[i + 1] * hrv2;
            const double p1 = d1[i] * hrv1, p2 = d2[i] * hrv2;
            const double n1 = d1[i + 2] * hrv1, n2 = d2[i + 2] * hrv2;
            bj = c0 + 0.5 * (p1 - n1) + (p2 - 2.0 * c2 + n2);
          } else {
            const double c0 = st[i] * hrv0, c2 = d2[i + 1] * r2(j);
            const double p1 = j > 0 ? d1[i] * r1(j - 1) : 0.0, p2 = j > 0 ? d2[i] * r2(j - 1) : 0.0;
            const double n1 = j + 1 < T ? d1[i + 2] * r1(j + 1) : 0.0, n2 = j + 1 < T ? d2[i + 2] * r2(j + 1) : 0.0;
            bj = c0 + 0.5 * (p1 - n1) + (p2 - 2.0 * c2 + n2);
          }
          base[i * RING_LANES] = hd < D ? bj : 0.0;
        }
      }
      ++mine;
      if (lane == 0) ring_post(prog + 1 + h, mine);          // (a wave's LDS operations execute in order: the segment is in the ring)
    };
    // ---- the same with 16-byte accesses: a lane owns TWO dimensions (2 L, 2 L + 1 of the block; L = lane & 31) and a
    // memory instruction covers two rows (lanes 0 .. 31 the even row of a pair, 32 .. 63 the odd one): half the
    // memory instructions for the same bytes.  Pays where the instructions, not the bytes, are what the way forward
    // waits for: float32 rows at many rounds of workgroups (the host chooses; see the launch).  The rows a frame needs
    // from the other half of the wave -- its neighbours -- come over with v_permlane32_swap.
    typedef double V2d __attribute__((ext_vector_type(2), aligned(8)));       // (rows are 8-byte aligned, not 16)
    typedef FT V2f __attribute__((ext_vector_type(2), aligned(sizeof(FT))));
    const int wh = lane >> 5, wl = lane & 31;
    const int wd0 = blk * RING_LANES + 2 * wl;                   // this lane's dimensions wd0, wd0 + 1 (D even: both live or neither)
    const bool wlive = wd0 < D;
    const int wdc = wlive ? wd0 : 0;
    V2d wrv0, wrv1, wrv2;
    const FT* whf = nullptr;
    double* wo = nullptr;
    if constexpr (WIDE) {
      wrv0 = V2d{1.0 / a.var[wdc], 1.0 / a.var[wdc + 1]};
      wrv1 = V2d{1.0 / a.var[D + wdc], 1.0 / a.var[D + wdc + 1]};
      wrv2 = V2d{1.0 / a.var[2 * D + wdc], 1.0 / a.var[2 * D + wdc + 1]};
      whf = (std::is_same<FT, float>::value ? reinterpret_cast<const FT*>(g.feat32) : reinterpret_cast<const FT*>(a.feat)) +
            t0 * a.ld_feat + a.col0 + wdc;
      wo = a.out + t0 * a.ld_out + a.ocol0 + wdc;
    }
    auto widen = [](V2f v) { return V2d{(double)v.x, (double)v.y}; };
    // the other half's value of x in this lane (lanes < 32 get what lanes >= 32 hold and the other way round), as the
    // pair (lower half's view, upper half's view) the selections below pick from
    auto swap1 = [](double x, double& from_upper, double& from_lower) {
      const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
      const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      // r[0]: lanes 32 .. 63 now hold the lower half's values; r[1]: lanes 0 .. 31 hold the upper half's
      from_lower = __hiloint2double((int)rh[0], (int)rl[0]);
      from_upper = __hiloint2double((int)rh[1], (int)rl[1]);
    };
    auto swap_halves = [&](V2d x, V2d& from_upper, V2d& from_lower) {
      double ux, uy, lx, ly;
      swap1(x.x, ux, lx);
      swap1(x.y, uy, ly);
      from_upper = V2d{ux, uy};
      from_lower = V2d{lx, ly};
    };
    auto forward_segment_wide = [&](int q, auto inner_tag) {
      constexpr bool INNER = decltype(inner_tag)::value;
      constexpr int NP = RING_SEG / 2;                 // row pairs of the segment: pair p = rows j0 + 2 p - 2, j0 + 2 p - 1
      const int j0 = q * RING_SEG;
      double* base = ring + seg_slot(q) * RING_LANES + 2 * wl;
      V2d st[NP], d1[NP + 2], d2[NP + 2];
#pragma unroll
      for (int p = 0; p < NP + 2; ++p) {
        int r = j0 - 2 + 2 * p + wh;
        if (!INNER) r = r < 0 ? 0 : (r < T ? r : T - 1);
        const FT* row = whf + (int64_t)r * a.ld_feat;
        d1[p] = widen(RING_LD(1, reinterpret_cast<const V2f*>(row + D)));
        d2[p] = widen(RING_LD(1, reinterpret_cast<const V2f*>(row + 2 * D)));
        if (p >= 1 && p <= NP) st[p - 1] = widen(RING_LD(1, reinterpret_cast<const V2f*>(row)));
      }
      while (q - ring_peek(prog) >= ring_segs) __builtin_amdgcn_s_sleep(2);      // the sweep has left segment q - ring_segs
      if (j0 >= RING_CAP && wlive) {                   // the y of frames j0 - RING_CAP .. leave the ring
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int i = 2 * k + wh;
          if (INNER || j0 + i < T)
            RING_ST(8, reinterpret_cast<V2d*>(wo + (int64_t)(j0 + i - RING_CAP) * a.ld_out), *reinterpret_cast<const V2d*>(base + i * RING_LANES));
        }
      }
      // u[p] (delta: u1, delta-delta: u2): the entries of row j0 + 2 p - 3 + wh -- for the frame of pair p in this lane
      // (row j0 + 2 p - 2 + wh) the row before it, for the frame of pair p - 1 the row after it.  Formed pair by pair
      // and used at once (all of them held would be 112 registers): frame k = p - 2 wants u[p - 1] and u[p].
      V2d up1 = V2d{0.0, 0.0}, up2 = up1;              // pair p - 1 as the lower half sees the upper one (its odd row)
      V2d uq1 = up1, uq2 = up1;                        // u[p - 1]
#pragma unroll
      for (int p = 0; p < NP + 2; ++p) {
        V2d fu1, fl1, fu2, fl2;
        swap_halves(d1[p], fu1, fl1);
        swap_halves(d2[p], fu2, fl2);
        const V2d uc1 = wh ? fl1 : up1, uc2 = wh ? fl2 : up2;      // u[p] (p >= 1)
        up1 = fu1;
        up2 = fu2;
        if (p >= 2) {
          const int k = p - 2, i = 2 * k + wh, j = j0 + i;
          if (INNER || j < T) {
            V2d bj;
            if (INNER) {
              const V2d c0 = st[k] * wrv0, c2 = d2[k + 1] * wrv2;
              const V2d p1 = uq1 * wrv1, p2 = uq2 * wrv2;
              const V2d n1 = uc1 * wrv1, n2 = uc2 * wrv2;
              bj = c0 + 0.5 * (p1 - n1) + (p2 - 2.0 * c2 + n2);
            } else {
              const V2d zero = V2d{0.0, 0.0}, edge = V2d{rvb, rvb};
              auto e1 = [&](int t) { return (t == 0 || t == T - 1) ? edge : wrv1; };
              auto e2 = [&](int t) { return (t == 0 || t == T - 1) ? edge : wrv2; };
              const V2d c0 = st[k] * wrv0, c2 = d2[k + 1] * e2(j);
              const V2d p1 = j > 0 ? uq1 * e1(j - 1) : zero, p2 = j > 0 ? uq2 * e2(j - 1) : zero;
              const V2d n1 = j + 1 < T ? uc1 * e1(j + 1) : zero, n2 = j + 1 < T ? uc2 * e2(j + 1) : zero;
              bj = c0 + 0.5 * (p1 - n1) + (p2 - 2.0 * c2 + n2);
            }
            *reinterpret_cast<V2d*>(base + i * RING_LANES) = wlive ? bj : V2d{0.0, 0.0};
          }
        }
        uq1 = uc1;
        uq2 = uc2;
      }
      ++mine;
      if (lane == 0) ring_post(prog + 1 + h, mine);
    };
    for (int q = h; q < nseg; q += RING_HELPERS) {
      const bool inner = q >= 1 && q * RING_SEG + RING_SEG <= T - 2;
      if constexpr (WIDE) {
        if (inner) forward_segment_wide(q, std::true_type{});
        else forward_segment_wide(q, std::false_type{});
      } else {
        if (inner) forward_segment(q, std::true_type{});
        else forward_segment(q, std::false_type{});
      }
    }
    // ---- backward: x of a finished segment out, then the y of a ring's length further down back into its slots
    int fetched = 0;
    // this wave's segments, highest first
    int qtop = nseg - 1;
    while (qtop >= 0 && qtop % RING_HELPERS != h) --qtop;
    while (ring_peek(prog + 8) == 0) __builtin_amdgcn_s_sleep(2);
    for (int q = qtop; q >= 0; q -= RING_HELPERS) {
      const int j0 = q * RING_SEG;
      double* base = ring + seg_slot(q) * RING_LANES + lane;
      const int qf = q - ring_segs;          // its frames went out on the way forward iff frame + RING_CAP < T
      // (the bytes read here were written in the forward phase, before prog[8] was posted -- no later store of this
      // workgroup touches them before this load -- so the loads need not wait for the sweep either)
      if constexpr (WIDE) {
        constexpr int NP = RING_SEG / 2;
        double* wbase = ring + seg_slot(q) * RING_LANES + 2 * wl;
        V2d yw[NP];
        if (qf >= 0 && wlive) {
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            const int j = qf * RING_SEG + 2 * k + wh;
            yw[k] = (j + RING_CAP < T) ? RING_LD(4, reinterpret_cast<const V2d*>(wo + (int64_t)j * a.ld_out)) : V2d{0.0, 0.0};
          }
        }
        while (ring_peek(prog + 16) > q) __builtin_amdgcn_s_sleep(2);     // the backward sweep has finished segment q
        if (wlive) {
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            const int i = 2 * k + wh;
            if (j0 + i < T) RING_ST(2, reinterpret_cast<V2d*>(wo + (int64_t)(j0 + i) * a.ld_out), *reinterpret_cast<const V2d*>(wbase + i * RING_LANES));
          }
          if (qf >= 0) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
              const int i = 2 * k + wh;
              if (qf * RING_SEG + i + RING_CAP < T) *reinterpret_cast<V2d*>(wbase + i * RING_LANES) = yw[k];
            }
          }
        }
      } else {
      double yv[RING_SEG];
      if (qf >= 0 && hd < D) {
#pragma unroll
        for (int k = 0; k < RING_SEG; ++k) {
          const int j = qf * RING_SEG + k;
          yv[k] = (j + RING_CAP < T) ? RING_LD(4, o + (int64_t)j * a.ld_out) : 0.0;
        }
      }
      while (ring_peek(prog + 16) > q) __builtin_amdgcn_s_sleep(2);     // the backward sweep has finished segment q
      if (hd < D) {
#pragma unroll
        for (int k = 0; k < RING_SEG; ++k)
          if (j0 + k < T) RING_ST(2, o + (int64_t)(j0 + k) * a.ld_out, base[k * RING_LANES]);
        if (qf >= 0) {
#pragma unroll
          for (int k = 0; k < RING_SEG; ++k) {
            const int j = qf * RING_SEG + k;
            if (j + RING_CAP < T) base[k * RING_LANES] = yv[k];
          }
        }
      }
      }
      ++fetched;
      if (lane == 0) ring_post(prog + 9 + h, fetched);
    }
    return;
  }

  // ================= wave 0: the two sweeps, a lane a dimension
  const double v0 = a.var[dc], v1 = a.var[D + dc], v2 = a.var[2 * D + dc];
  const double tau0 = 1.0 / v0, tau1_in = 1.0 / v1, tau2_in = 1.0 / v2, tau_edge = 1.0 / kBigVar;
  auto tau1 = [&](int t) -> double {
    if (t < 0 || t >= T) return 0.0;
    return (t == 0 || t == T - 1) ? tau_edge : tau1_in;
  };
  auto tau2 = [&](int t) -> double {
    if (t < 0 || t >= T) return 0.0;
    return (t == 0 || t == T - 1) ? tau_edge : tau2_in;
  };
  // the shared factor's view of the variances ("T = infinity": an edge at frame 0 only), as mlpg_factor_block has it
  auto tau1f = [&](int t) -> double { return t < 0 ? 0.0 : (t == 0 ? tau_edge : tau1_in); };
  auto tau2f = [&](int t) -> double { return t < 0 ? 0.0 : (t == 0 ? tau_edge : tau2_in); };
  double* fd = a.scratch + dc;
  double* fl1 = fd + plane;
  double* fl2 = fl1 + plane;
  const double pjj_in = tau0 + 0.25 * (tau1f(1) + tau1f(3)) + (tau2f(1) + 4.0 * tau2f(2) + tau2f(3));
  const double pj1_in = -2.0 * (tau2f(2) + tau2f(3));
  const double pj2_in = tau2f(1) - 0.25 * tau1f(1);          // (row 0's too)
  const int n_shared = T >= 3 ? T - 2 : 0;
  // the factor: rows 0 .. ncvmax derived on the way forward (a lane's entries stay put from its own row of repetition
  // on: mlpg_factor_block's rule), the stationary entries in three registers from there
  double sd = 0.0, sl1 = 0.0, sl2 = 0.0;
  bool lane_settled = false;             // this lane's factor has repeated: (sd, sl1, sl2) hold
  bool settled = false;                  // every lane's has
  int ncvmax = 0x7fffffff;               // the row at which the last lane's did
  double tl_d0 = 1.0, tl_d1 = 1.0, tl_10 = 0.0, tl_11 = 0.0, tl_20 = 0.0, tl_21 = 0.0;      // frames T - 2, T - 1
  double l1p = 0.0, l2p = 0.0, cprev = 0.0, y1 = 0.0, y2 = 0.0;
  double* lane_ring = ring + lane;

  __builtin_amdgcn_s_setprio(3);         // (the SIMD is shared with a helper wave: the sweep goes first)
  auto relaxed = [](const int* w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

  // ---- forward
  int hsel = 0, hcnt = 0, slot = 0;       // segment sgm: helper sgm % H, its (sgm / H + 1)-th; first ring slot
  int seen = relaxed(prog + 1);
  for (int sgm = 0; sgm < nseg; ++sgm) {
    // (what the sweep needs to know about the segment after this one -- is it in the ring yet? -- is asked for before
    // the chain and looked at after it)
    if (seen <= hcnt)
      while (ring_peek(prog + 1 + hsel) <= hcnt) __builtin_amdgcn_s_sleep(1);   // segment sgm is in the ring
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int hsel_n = hsel + 1 == RING_HELPERS ? 0 : hsel + 1, hcnt_n = hsel + 1 == RING_HELPERS ? hcnt + 1 : hcnt;
    seen = relaxed(prog + 1 + hsel_n);
    const int j0 = sgm * RING_SEG;
    const int jend = j0 + RING_SEG < T ? j0 + RING_SEG : T;
    double* sl = lane_ring + slot * RING_LANES;         // slot of frame j: sl[(j - j0) * 64]
    if (settled && j0 >= ncvmax + 2 && j0 + RING_SEG <= n_shared) {
      // the whole segment on the stationary factor, two stationary frames behind it: straight-line code, the 24
      // right-hand sides read at once, the chain, the writes
      double v[RING_SEG];
#pragma unroll
      for (int i = 0; i < RING_SEG; ++i) v[i] = sl[i * RING_LANES];
#pragma unroll
      for (int i = 0; i < RING_SEG; ++i) {
        const double y = (v[i] - sl1 * y1 - sl2 * y2) * sd;
        sl[i * RING_LANES] = y;
        y2 = y1;
        y1 = y;
      }
    } else {
      // the head (the factor still moves: derived here, row by row, and left in the table for the way back), the
      // frames between it and the first whole stationary segment, the segment with the two re-derived tail frames:
      // a rolled loop, the right-hand side of the frame after asked for first
      double nb = sl[0];
#pragma unroll 1
      for (int j = j0; j < jend; ++j) {
        const int jn = j + 1 < jend ? j + 1 : j;
        const double nb_n = sl[(jn - j0) * RING_LANES];
        double dd = sd, l1 = sl1, l2 = sl2;                    // dd holds 1 / L[j,j]
        if (j >= n_shared) {
          const double pjj = tau0 + 0.25 * (tau1(j - 1) + tau1(j + 1)) + (tau2(j - 1) + 4.0 * tau2(j) + tau2(j + 1));
          const double pj1 = (j + 1 < T) ? -2.0 * (tau2(j) + tau2(j + 1)) : 0.0;
          const double pj2 = (j + 2 < T) ? (tau2(j + 1) - 0.25 * tau1(j + 1)) : 0.0;
          dd = 1.0 / sqrt(pjj - l1p * l1p - l2p * l2p);
          l1 = (pj1 - cprev * l1p) * dd;
          l2 = pj2 * dd;
          const int q = j - (T - 2);
          if (q == 0) { tl_d0 = dd; tl_10 = l1; tl_20 = l2; }
          else { tl_d1 = dd; tl_11 = l1; tl_21 = l2; }
        } else if (!settled) {
          if (!lane_settled) {
            // (P's entries are those of row 2 from there on: the same expressions on the same values)
            double pjj = pjj_in, pj1 = pj1_in;
            if (j < 2) {
              pjj = tau0 + 0.25 * (tau1f(j - 1) + tau1f(j + 1)) + (tau2f(j - 1) + 4.0 * tau2f(j) + tau2f(j + 1));
              pj1 = -2.0 * (tau2f(j) + tau2f(j + 1));
            }
            dd = factor_rsqrt(pjj - l1p * l1p - l2p * l2p);
            l1 = (pj1 - cprev * l1p) * dd;
            l2 = pj2_in * dd;
            auto same = [](double x, double y) { return fabs(x - y) <= 8.9e-16 * fabs(y); };
            if (j >= 3 && same(l1, l1p) && same(l2, cprev) && same(cprev, l2p)) {
              lane_settled = true;
              sd = dd; sl1 = l1; sl2 = l2;
            }
          }
          fd[(int64_t)j * D] = dd;           // (every workgroup leaves the same values here)
          fl1[(int64_t)j * D] = l1;
          fl2[(int64_t)j * D] = l2;
          if (__all(lane_settled)) {
            settled = true;
            ncvmax = j;
          }
        }
        const double y = (nb - l1p * y1 - l2p * y2) * dd;
        sl[(j - j0) * RING_LANES] = y;
        l2p = cprev;
        l1p = l1;
        cprev = l2;
        y2 = y1;
        y1 = y;
        nb = nb_n;
      }
    }
    if (lane == 0) ring_post(prog, sgm + 1);
    hsel = hsel_n; hcnt = hcnt_n; slot = slot + RING_SEG == RING_CAP ? 0 : slot + RING_SEG;
  }
  // the helpers have prepared everything (the sweep consumed it); their counters start again for the way back
  // ---- backward: L^T x = y, segments from the last to the first
  // the head's rows: 0 .. ncvmax - 1 where the factor settled (row ncvmax on is the registers'), else every shared row
  const int head_rows = settled ? ncvmax : n_shared;
  const int head_last = settled ? ncvmax : n_shared - 1;      // the last row of the table (settled: the stationary one)
  if (lane == 0) {
    ring_post(prog + 16, nseg);          // lowest finished segment: none yet
    ring_post(prog + 8, 1);
  }
  double x1 = 0.0, x2 = 0.0;
  // segment sgm's y is still in the ring, or comes back with the helper that stores segment qs = sgm + ring_segs: helper
  // qs % H, whose count stands at (nseg - 1 - qs) / H + 1 after that segment (it takes its segments from the top)
  slot = ((nseg - 1) % ring_segs) * RING_SEG;
  int bq = 0, bh = 0, bneed = 0;           // for the segment at hand: bq >= 0: it has to wait, for helper bh to count bneed
  auto counters_for = [&](int sgm) {
    const int qs = sgm + ring_segs;
    bh = qs % RING_HELPERS;
    bq = nseg - 1 - qs;
    bneed = bq >= 0 ? bq / RING_HELPERS + 1 : 0;
  };
  counters_for(nseg - 1);
  int seen_b = bq >= 0 ? relaxed(prog + 9 + bh) : 0;
  // The head's rows come back from the table -- this workgroup's own rows of it, a sweep's length old.  A trip to the
  // L2 is 0.5 - 1 us, five to ten frames of this sweep: a segment that reaches into the head (and holds none of the
  // two tail frames) is therefore straight-line code in two halves of twelve rows, the rows of a half (1 / L[j,j] and
  // L[j+1,j]; L[j+2,j] is pj2 times the first, as it was formed) asked for while the half before it is worked; frames
  // of it above the head read the table's last row, which holds the stationary entries.  (Until late in round 5 the
  // helpers put heads of up to 56 rows into free slots of the ring for a rolled loop to read: slower than this for 50
  // rows -- 2.84 against 2.76 ms at 4 096 utterances -- and a progress protocol of its own.)
  constexpr int HALF = RING_SEG / 2;
  double ud[HALF], u1[HALF], wd[HALF], w1[HALF];        // upper half (rows j0 + 23 .. j0 + 12), lower half (j0 + 11 .. j0)
  auto table_seg = [&](int sg) {
    return sg >= 0 && sg * RING_SEG < head_rows && sg * RING_SEG + RING_SEG <= n_shared;
  };
  auto load_upper = [&](int sg) {
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      const int row = sg * RING_SEG + RING_SEG - 1 - i;
      const int64_t r = (int64_t)(row < head_last ? row : head_last) * D;
      ud[i] = fd[r];
      u1[i] = fl1[r];
    }
  };
  auto load_lower = [&](int sg) {
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      const int row = sg * RING_SEG + HALF - 1 - i;
      const int64_t r = (int64_t)(row < head_last ? row : head_last) * D;
      wd[i] = fd[r];
      w1[i] = fl1[r];
    }
  };
  for (int sgm = nseg - 1; sgm >= 0; --sgm) {
    if (bq >= 0 && seen_b < bneed)
      while (ring_peek(prog + 9 + bh) < bneed) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (sgm > 0) {
      counters_for(sgm - 1);
      seen_b = bq >= 0 ? relaxed(prog + 9 + bh) : 0;
    }
    const int j0 = sgm * RING_SEG;
    const int jtop = (j0 + RING_SEG < T ? j0 + RING_SEG : T) - 1;
    double* sl = lane_ring + slot * RING_LANES;
    const bool cur_tab = table_seg(sgm), next_tab = table_seg(sgm - 1);
    if (!cur_tab && next_tab) load_upper(sgm - 1);
    if (cur_tab) {
      load_lower(sgm);
      double v[HALF];
#pragma unroll
      for (int i = 0; i < HALF; ++i) v[i] = sl[(RING_SEG - 1 - i) * RING_LANES];
#pragma unroll
      for (int i = 0; i < HALF; ++i) {
        const double x = (v[i] - u1[i] * x1 - (pj2_in * ud[i]) * x2) * ud[i];
        sl[(RING_SEG - 1 - i) * RING_LANES] = x;
        x2 = x1;
        x1 = x;
      }
      if (next_tab) load_upper(sgm - 1);
#pragma unroll
      for (int i = 0; i < HALF; ++i) v[i] = sl[(HALF - 1 - i) * RING_LANES];
#pragma unroll
      for (int i = 0; i < HALF; ++i) {
        const double x = (v[i] - w1[i] * x1 - (pj2_in * wd[i]) * x2) * wd[i];
        sl[(HALF - 1 - i) * RING_LANES] = x;
        x2 = x1;
        x1 = x;
      }
    } else if (j0 >= head_rows && j0 + RING_SEG <= n_shared) {
      double v[RING_SEG];
#pragma unroll
      for (int i = 0; i < RING_SEG; ++i) v[i] = sl[(RING_SEG - 1 - i) * RING_LANES];
#pragma unroll
      for (int i = 0; i < RING_SEG; ++i) {
        const double x = (v[i] - sl1 * x1 - sl2 * x2) * sd;
        sl[(RING_SEG - 1 - i) * RING_LANES] = x;
        x2 = x1;
        x1 = x;
      }
    } else {
      // rolled, as on the way forward (the segment with the tail frames; head rows in it -- an utterance shorter than
      // the head + a segment -- from the table, a row ahead)
      auto fetch = [&](int j, double& qd, double& q1, double& q2) {
        qd = sd; q1 = sl1; q2 = sl2;
        if (j >= n_shared) {
          const bool last = j == T - 1;
          qd = last ? tl_d1 : tl_d0; q1 = last ? tl_11 : tl_10; q2 = last ? tl_21 : tl_20;
        } else if (j < head_rows) {
          qd = fd[(int64_t)j * D]; q1 = fl1[(int64_t)j * D]; q2 = fl2[(int64_t)j * D];
        }
      };
      double ny = sl[(jtop - j0) * RING_LANES], cd, c1, c2;
      fetch(jtop, cd, c1, c2);
#pragma unroll 1
      for (int j = jtop; j >= j0; --j) {
        const int jn = j - 1 >= j0 ? j - 1 : j;
        const double ny_n = sl[(jn - j0) * RING_LANES];
        double nd, n1, n2;
        fetch(jn, nd, n1, n2);
        const double x = (ny - c1 * x1 - c2 * x2) * cd;
        sl[(j - j0) * RING_LANES] = x;
        x2 = x1;
        x1 = x;
        ny = ny_n; cd = nd; c1 = n1; c2 = n2;
      }
    }
    if (lane == 0) ring_post(prog + 16, sgm);
    slot = slot == 0 ? RING_CAP - RING_SEG : slot - RING_SEG;
  }
}

#undef RING_LD
#undef RING_ST

// ---- chunk geometry, factor access and the two sweeps of one chunk -------------------------------
// (shared by the reduce and the solve kernel below)
template <int FU_FL>
__device__ __host__ __forceinline__ int fu_num_chunks(int64_t T) { return (int)((T + FU_FL - 1) / FU_FL); }
// the last chunk always holds both re-derived tail frames: a one-frame remainder takes a frame
// from the chunk before it
template <int FU_FL>
__device__ __forceinline__ int64_t fu_chunk_start(int k, int K, int64_t T) {
  int64_t s = (int64_t)k * FU_FL;
  if (k == K - 1 && K > 1 && T - s == 1) s -= 1;
  return k >= K ? T : s;
}

struct FuFac {
  const double* fd;
  const double* fl1;
  const double* fl2;
  int64_t ncv, n_shared, T;
  int D;
  double tau0, tau1_in, tau2_in;
  __device__ __forceinline__ double F(const double* pl, int64_t j) const {
    return j < 0 ? 0.0 : pl[(j < ncv ? j : ncv) * D];
  }
  __device__ __forceinline__ double tau1(int64_t t) const {
    if (t < 0 || t >= T) return 0.0;
    return (t == 0 || t == T - 1) ? 1.0 / kBigVar : tau1_in;
  }
  __device__ __forceinline__ double tau2(int64_t t) const {
    if (t < 0 || t >= T) return 0.0;
    return (t == 0 || t == T - 1) ? 1.0 / kBigVar : tau2_in;
  }
  // factor of a tail frame (j >= n_shared) from the Cholesky state that reaches it
  __device__ __forceinline__ void derive(int64_t j, double l1p, double l2p, double cprev, double& dd,
                                         double& l1, double& l2) const {
    const double pjj = tau0 + 0.25 * (tau1(j - 1) + tau1(j + 1)) + (tau2(j - 1) + 4.0 * tau2(j) + tau2(j + 1));
    const double pj1 = (j + 1 < T) ? -2.0 * (tau2(j) + tau2(j + 1)) : 0.0;
    const double pj2 = (j + 2 < T) ? (tau2(j + 1) - 0.25 * tau1(j + 1)) : 0.0;
    dd = 1.0 / sqrt(pjj - l1p * l1p - l2p * l2p);
    l1 = (pj1 - cprev * l1p) * dd;
    l2 = pj2 * dd;
  }
};

// how many of the chunk's n frames starting at j0 take the shared factor (the rest -- at most the
// utterance's last two -- are re-derived)
__device__ __forceinline__ int fu_shared_frames(const FuFac& c, int64_t j0, int n) {
  const int64_t m = c.n_shared - j0;
  return m <= 0 ? 0 : (m < n ? (int)m : n);
}

// Forward sweep over the chunk [j0, j0 + n).  PASS_A: zero-state response e and the two unit
// responses M, b untouched; else: from (s1, s2), y replaces b.  tl: factors of frames T-2, T-1.
template <int FU_FL, bool CONST, bool PASS_A>
__device__ __forceinline__ void fu_fwd(const FuFac& c, double (&b)[FU_FL], int64_t j0, int n, double s1,
                                       double s2, double (&M)[4], double (&e)[2], double (&tl)[6]) {
  double kd = 0.0, k1 = 0.0, k2 = 0.0;
  double l1p, l2p, cprev;
  if (CONST) {
    kd = c.fd[c.ncv * c.D];
    k1 = c.fl1[c.ncv * c.D];
    k2 = c.fl2[c.ncv * c.D];
    l1p = k1; l2p = k2; cprev = k2;
  } else {
    l1p = c.F(c.fl1, j0 - 1); l2p = c.F(c.fl2, j0 - 2); cprev = c.F(c.fl2, j0 - 1);
    if (c.n_shared == 0) l1p = l2p = cprev = 0.0;
  }
  double y1 = s1, y2 = s2, u1 = 1.0, u2 = 0.0, v1 = 0.0, v2 = 1.0;
  // frames that take the shared factor first (unrolled), then the utterance's last two, whose
  // factor is re-derived (a rolled loop: one copy of the square root and divisions in the code)
  const int n_main = CONST ? FU_FL : fu_shared_frames(c, j0, n);
#pragma unroll
  for (int i = 0; i < FU_FL; ++i) {
    if (CONST || i < n_main) {
      const int64_t j = j0 + i;
      double dd, l1, l2;
      if (CONST) {
        dd = kd; l1 = k1; l2 = k2;
      } else {
        const int64_t jc = (j < c.ncv ? j : c.ncv) * c.D;
        dd = c.fd[jc]; l1 = c.fl1[jc]; l2 = c.fl2[jc];
      }
      const double y = (b[i] - l1p * y1 - l2p * y2) * dd;
      if (PASS_A) {
        const double u = (-l1p * u1 - l2p * u2) * dd;
        const double v = (-l1p * v1 - l2p * v2) * dd;
        u2 = u1; u1 = u; v2 = v1; v1 = v;
      } else {
        b[i] = y;
      }
      y2 = y1; y1 = y;
      l2p = cprev; l1p = l1; cprev = l2;
    }
  }
  if (!CONST) {
#pragma unroll 1
    for (int i = n_main; i < n; ++i) {
      const int64_t j = j0 + i;
      double dd, l1, l2;
      c.derive(j, l1p, l2p, cprev, dd, l1, l2);
      if (j == c.T - 1) { tl[3] = dd; tl[4] = l1; tl[5] = l2; }
      else { tl[0] = dd; tl[1] = l1; tl[2] = l2; }
      double bi = 0.0;
#pragma unroll
      for (int r = 0; r < FU_FL; ++r) bi = r == i ? b[r] : bi;
      const double y = (bi - l1p * y1 - l2p * y2) * dd;
      if (PASS_A) {
        const double u = (-l1p * u1 - l2p * u2) * dd;
        const double v = (-l1p * v1 - l2p * v2) * dd;
        u2 = u1; u1 = u; v2 = v1; v1 = v;
      } else {
#pragma unroll
        for (int r = 0; r < FU_FL; ++r) b[r] = r == i ? y : b[r];
      }
      y2 = y1; y1 = y;
      l2p = cprev; l1p = l1; cprev = l2;
    }
  }
  if (PASS_A) {
    M[0] = u1; M[1] = v1; M[2] = u2; M[3] = v2;
    e[0] = y1; e[1] = y2;
  }
}

// Backward sweep over y (in b).  PASS_A: (M, e) of the chunk; else: x from (s1, s2) = (x_{j1},
// x_{j1+1}) written to `o` (row pitch ldo), when `store`.
template <int FU_FL, bool CONST, bool PASS_A>
__device__ __forceinline__ void fu_bwd(const FuFac& c, double (&b)[FU_FL], int64_t j0, int n, double s1,
                                       double s2, double (&M)[4], double (&e)[2], const double (&tl)[6],
                                       double* o, int64_t ldo, bool store) {
  double kd = 0.0, k1 = 0.0, k2 = 0.0;
  if (CONST) {
    kd = c.fd[c.ncv * c.D];
    k1 = c.fl1[c.ncv * c.D];
    k2 = c.fl2[c.ncv * c.D];
  }
  double x1 = s1, x2 = s2, u1 = 1.0, u2 = 0.0, v1 = 0.0, v2 = 1.0;
#pragma unroll
  for (int i = FU_FL - 1; i >= 0; --i) {
    if (CONST || i < n) {
      const int64_t j = j0 + i;
      double dd, l1, l2;
      if (CONST) {
        dd = kd; l1 = k1; l2 = k2;
      } else if (j < c.n_shared) {
        const int64_t jc = (j < c.ncv ? j : c.ncv) * c.D;
        dd = c.fd[jc]; l1 = c.fl1[jc]; l2 = c.fl2[jc];
      } else {
        const bool last = j == c.T - 1;
        dd = last ? tl[3] : tl[0]; l1 = last ? tl[4] : tl[1]; l2 = last ? tl[5] : tl[2];
      }
      const double x = (b[i] - l1 * x1 - l2 * x2) * dd;
      if (PASS_A) {
        const double u = (-l1 * u1 - l2 * u2) * dd;
        const double v = (-l1 * v1 - l2 * v2) * dd;
        u2 = u1; u1 = u; v2 = v1; v1 = v;
      } else if (store) {
        o[j * ldo] = x;
      }
      x2 = x1; x1 = x;
    }
  }
  if (PASS_A) {
    M[0] = u1; M[1] = v1; M[2] = u2; M[3] = v2;
    e[0] = x1; e[1] = x2;
  }
}

// b = W^T (mean / var) of the frames [j0, j0 + n) of one utterance (mlpg.py:123) for this lane's
// dimension; f: the lane's column in the utterance's first row.  `interior`: rows j0-1 .. j0+FL
// all exist and none is an edge frame (no clamping, no edge variances).
template <int FU_FL>
__device__ __forceinline__ void fu_form_b(const MlpgArgs& a, const double* f, int64_t j0, int n, int64_t T,
                                          bool interior, double v0, double v1, double v2,
                                          double (&b)[FU_FL]) {
  const int D = a.dim;
  const bool cst = interior;
  const double rv0 = 1.0 / v0, rv1 = 1.0 / v1, rv2 = 1.0 / v2, rvb = 1.0 / kBigVar;
  if (cst) {
    // interior chunk: rows j0-1 .. j0+FU_FL all exist and none is an edge frame
    const double* r0 = f + j0 * a.ld_feat;
#pragma unroll
    for (int i = 0; i < FU_FL; ++i) b[i] = r0[(int64_t)i * a.ld_feat] * rv0;
    {
      double m2 = 0.0, m1 = 0.0;
#pragma unroll
      for (int i = -1; i <= FU_FL; ++i) {
        const double v = r0[(int64_t)i * a.ld_feat + D] * rv1;
        if (i >= 1) b[i - 1] += 0.5 * (m2 - v);
        m2 = m1; m1 = v;
      }
    }
    {
      double m2 = 0.0, m1 = 0.0;
#pragma unroll
      for (int i = -1; i <= FU_FL; ++i) {
        const double v = r0[(int64_t)i * a.ld_feat + 2 * D] * rv2;
        if (i >= 1) b[i - 1] += (m2 - 2.0 * m1 + v);
        m2 = m1; m1 = v;
      }
    }
  } else {
    auto rowp = [&](int64_t r) { return f + (r < 0 ? 0 : (r >= T ? T - 1 : r)) * a.ld_feat; };
#pragma unroll
    for (int i = 0; i < FU_FL; ++i) b[i] = i < n ? rowp(j0 + i)[0] * rv0 : 0.0;
    {
      double m2 = 0.0, m1 = 0.0;
#pragma unroll
      for (int i = -1; i <= FU_FL; ++i) {
        const int64_t r = j0 + i;
        double v = 0.0;
        if (i <= n && r >= 0 && r < T) v = rowp(r)[D] * ((r == 0 || r == T - 1) ? rvb : rv1);
        if (i >= 1 && i - 1 < n) b[i - 1] += 0.5 * (m2 - v);
        m2 = m1; m1 = v;
      }
    }
    {
      double m2 = 0.0, m1 = 0.0;
#pragma unroll
      for (int i = -1; i <= FU_FL; ++i) {
        const int64_t r = j0 + i;
        double v = 0.0;
        if (i <= n && r >= 0 && r < T) v = rowp(r)[2 * D] * ((r == 0 || r == T - 1) ? rvb : rv2);
        if (i >= 1 && i - 1 < n) b[i - 1] += (m2 - 2.0 * m1 + v);
        m2 = m1; m1 = v;
      }
    }
  }
}

// ---- dependency-free solve: reduce -> scan -> solve ----------------------------------------------
// The fused kernel above reads the input once, but its workgroups spend two thirds of their life
// waiting for each other (every wait ends with the slowest load among the waves it depends on)
// while their registers hold the chunk, and the registers bound how much of the batch is in
// flight: 15-19 % of the HBM peak at any batch size.  This form has no
// wait at all.  It rests on two facts: the forward sweep is linear in (b, entry state), and the
// chunk-local backward sweep x = L_cc^-T y has the adjoint form x_0 = (L_cc^-1 e_0) . y,
// x_1 = (L_cc^-1 e_1) . y -- so what the backward sweep of a chunk contributes to the chunk in
// front of it can be accumulated WHILE WALKING FORWARD, as two dot products with the forward
// impulse responses P = L_cc^-1 e_0 and R = L_cc^-1 e_1, without keeping y:
//   reduce  every chunk, from a zero entry state: e_f = (y0_{n-1}, y0_{n-2}), e_b0 = (P.y0, R.y0);
//           one streaming read of the input, four doubles out per (chunk, dimension), no state
//   scan    per (utterance, dimension): s_in(k+1) = M_f s_in(k) + e_f(k), then backwards
//           t_in(k-1) = M_b t_in(k) + e_b0(k) + C s_in(k); the matrices are data-independent --
//           M_f = the entry state's image (a combination of the last two P, R), M_b = the exit
//           state's image (P, R at the last two frames times the factor's off-diagonals),
//           C = [P R]^T [U V] from the Gram sums P.P, P.R, R.R -- one set per dimension for the
//           stationary chunks, recomputed in place for the few others (utterance start / tail)
//   solve   every chunk again, now from its true states: b read back from the output rows (where the
//           reduce kernel left it), y in registers, x stored over b
// HBM bytes per frame: 1496 (input once) + 3 x 496 (b out, b in, x out) + the aggregates (128 B per
// chunk and dimension, written and read once each) = 3.1 kB against 2000 algorithmic; measured with the
// halo rows and partial lines 4.4 kB (profiles/r4_section_traffic.json).
struct alignas(32) StRecord {
  long long t0;      // first frame of the utterance in the batch
  int T;             // its length
  int k0;            // first chunk of this group (index inside the utterance)
  int chunk;         // batch-wide index of that chunk
  int pad[3];
};

struct StreamArgs {
  MlpgArgs a;
  int t_max;
  const StRecord* rec;   // [n_groups] groups of ST_GW consecutive chunks of one utterance
  const int* chunk0;     // [U+1] batch-wide index of every utterance's first chunk
  int n_groups, nblk;
  double* agg;           // [n_chunks][4][Dp]: e_f (2), e_b0 (2)
  double* st;            // [n_chunks][4][Dp]: forward entry state (2), backward entry state (2)
};


struct FuMats { double Mf[4], Mb[4], C[4]; };

// One forward walk over the chunk [j0, j0 + n) from a zero entry state, nothing kept.
// DATA: e = (y0_{n-1}, y0_{n-2}, P.y0, R.y0).  MATS: the chunk's data-independent matrices.
// PRELOAD (with !CONST): the chunk's factor rows are requested together before the walk instead of
// inside its (lane-divergent) branches -- one trip to memory per chunk instead of one per frame.
template <int FU_FL, bool CONST, bool DATA, bool MATS, bool PRELOAD = false>
__device__ __forceinline__ void fu_reduce(const FuFac& c, const double (&b)[FU_FL], int64_t j0, int n,
                                          double (&tl)[6], double (&e)[4], FuMats& m) {
  double kd = 0.0, k1 = 0.0, k2 = 0.0;
  double l1p, l2p, cprev;
  // (in two halves: the scan kernel that uses this runs sixteen waves per workgroup, 128 registers each,
  // and 3 x 16 doubles of factor rows on top of the walk's state spilled -- 836 bytes of scratch per lane,
  // 14-17 us for the matrices of ONE chunk; two trips to memory instead of one, no spill)
  constexpr int PH = PRELOAD ? FU_FL / 2 : 1;
  double pd[PH], p1[PH], p2[PH];
  auto preload = [&](int h) {
#pragma unroll
    for (int i = 0; i < PH; ++i) {
      const int64_t j = j0 + h * PH + i;
      const int64_t jc = (j < c.ncv ? j : c.ncv) * c.D;      // always a valid row of the factor
      pd[i] = c.fd[jc]; p1[i] = c.fl1[jc]; p2[i] = c.fl2[jc];
    }
  };
  if (PRELOAD && !CONST) preload(0);
  if (CONST) {
    kd = c.fd[c.ncv * c.D];
    k1 = c.fl1[c.ncv * c.D];
    k2 = c.fl2[c.ncv * c.D];
    l1p = k1; l2p = k2; cprev = k2;
  } else {
    l1p = c.F(c.fl1, j0 - 1); l2p = c.F(c.fl2, j0 - 2); cprev = c.F(c.fl2, j0 - 1);
    if (c.n_shared == 0) l1p = l2p = cprev = 0.0;
  }
  double y1 = 0.0, y2 = 0.0, P1 = 0.0, P2 = 0.0, R1 = 0.0, R2 = 0.0;
  double spy = 0.0, sry = 0.0, spp = 0.0, spr = 0.0, srr = 0.0;
  double rho0u = 0.0, rho0v = 0.0, rho1u = 0.0;      // what the entry state adds to frames 0 and 1
  double l1_last = 0.0, l2_last = 0.0, l2_prev = 0.0;  // own factor entries of frames n-1 and n-2
  const int n_main = CONST ? FU_FL : fu_shared_frames(c, j0, n);
  // one step of the walk; `first` / `second`: frame 0 / 1 of the chunk
  auto step = [&](bool first, bool second, double dd, double l1, double l2, double bi) {
    if (first) { rho0u = -l1p; rho0v = -l2p; }
    if (second) rho1u = -l2p;
    const double P = ((first ? 1.0 : 0.0) - l1p * P1 - l2p * P2) * dd;
    const double R = ((second ? 1.0 : 0.0) - l1p * R1 - l2p * R2) * dd;
    if (DATA) {
      const double y = (bi - l1p * y1 - l2p * y2) * dd;
      spy += P * y; sry += R * y;
      y2 = y1; y1 = y;
    }
    if (MATS) { spp += P * P; spr += P * R; srr += R * R; }
    P2 = P1; P1 = P; R2 = R1; R1 = R;
    l2_prev = l2_last; l1_last = l1; l2_last = l2;
    l2p = cprev; l1p = l1; cprev = l2;
  };
#pragma unroll
  for (int i = 0; i < FU_FL; ++i) {
    if (PRELOAD && !CONST && i == PH) preload(1);
    if (CONST || i < n_main) {
      const int64_t j = j0 + i;
      double dd, l1, l2;
      if (CONST) {
        dd = kd; l1 = k1; l2 = k2;
      } else if (PRELOAD) {
        dd = pd[i % PH]; l1 = p1[i % PH]; l2 = p2[i % PH];
      } else {
        const int64_t jc = (j < c.ncv ? j : c.ncv) * c.D;
        dd = c.fd[jc]; l1 = c.fl1[jc]; l2 = c.fl2[jc];
      }
      step(i == 0, i == 1, dd, l1, l2, DATA ? b[i] : 0.0);
    }
  }
  if (!CONST) {      // the utterance's last two frames: factor re-derived (rolled: one copy)
#pragma unroll 1
    for (int i = n_main; i < n; ++i) {
      const int64_t j = j0 + i;
      double dd, l1, l2;
      c.derive(j, l1p, l2p, cprev, dd, l1, l2);
      if (j == c.T - 1) { tl[3] = dd; tl[4] = l1; tl[5] = l2; }
      else { tl[0] = dd; tl[1] = l1; tl[2] = l2; }
      double bi = 0.0;
      if (DATA) {
#pragma unroll
        for (int r = 0; r < FU_FL; ++r) bi = r == i ? b[r] : bi;
      }
      step(i == 0, i == 1, dd, l1, l2, bi);
    }
  }
  if (DATA) { e[0] = y1; e[1] = y2; e[2] = spy; e[3] = sry; }
  if (MATS) {
    m.Mf[0] = rho0u * P1 + rho1u * R1; m.Mf[1] = rho0v * P1;
    m.Mf[2] = rho0u * P2 + rho1u * R2; m.Mf[3] = rho0v * P2;
    m.Mb[0] = -l1_last * P1 - l2_prev * P2; m.Mb[1] = -l2_last * P1;
    m.Mb[2] = -l1_last * R1 - l2_prev * R2; m.Mb[3] = -l2_last * R1;
    m.C[0] = rho0u * spp + rho1u * spr; m.C[1] = rho0v * spp;
    m.C[2] = rho0u * spr + rho1u * srr; m.C[3] = rho0v * spr;
  }
}

// what the reduce and the solve kernel share: which chunk this wave owns, its lane's constants
template <int FU_FL>
struct StChunk {
  int64_t t0, T, j0, j1;
  int K, k, n, chunk, d;
  bool dok, cst;
  double v0, v1, v2;
  FuFac c;
  __device__ __forceinline__ bool open(const StreamArgs& g) {
    const MlpgArgs& a = g.a;
    const int grp = (int)(blockIdx.x / (unsigned)g.nblk), db = (int)(blockIdx.x % (unsigned)g.nblk);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const StRecord rec = g.rec[grp];
    t0 = rec.t0; T = rec.T;
    K = fu_num_chunks<FU_FL>(T);
    k = rec.k0 + w;
    const bool active = k < K;
    if (!active) k = K - 1;          // an idle wave computes on the last chunk's geometry and stores nothing
    chunk = rec.chunk + w;
    const int D = a.dim;
    dok = db * 64 + lane < D;
    d = dok ? db * 64 + lane : D - 1;
    j0 = fu_chunk_start<FU_FL>(k, K, T); j1 = fu_chunk_start<FU_FL>(k + 1, K, T);
    n = (int)(j1 - j0);
    v0 = a.var[d]; v1 = a.var[D + d]; v2 = a.var[2 * D + d];
    const int64_t plane = (int64_t)g.t_max * D;
    c.fd = a.scratch + d; c.fl1 = c.fd + plane; c.fl2 = c.fl1 + plane;
    c.ncv = a.nconv[d]; c.n_shared = T >= 3 ? T - 2 : 0; c.T = T; c.D = D;
    c.tau0 = 1.0 / v0; c.tau1_in = 1.0 / v1; c.tau2_in = 1.0 / v2;
    const bool cst_lane = (j0 - 2 >= c.ncv) && (j1 <= c.n_shared) && n == FU_FL;
    cst = __all(cst_lane);
    return active;
  }
};

// Stages the rows [jlo, jhi) of one utterance -- the three 64-column pieces (static, delta,
// delta-delta) of this workgroup's dimension block -- into LDS as tile[row][w * 64 + lane], with
// every thread of the workgroup loading: the pieces of a row are contiguous in memory, so the loads
// are full-width (16 bytes per lane when the row pitch, the first column and the dimension count are
// even) and a row that two neighbouring chunks need is fetched once.  The waves then form b from
// LDS through fu_form_b with pitch ST_W.
constexpr int ST_W = 192;      // doubles per staged row

template <int NTHR, int MAXR>
__device__ __forceinline__ void st_stage_rows(const MlpgArgs& a, int db, int64_t t0, int64_t jlo, int rows,
                                              double* tile) {
  const int D = a.dim;
  const int dblk = D - db * 64 < 64 ? D - db * 64 : 64;
  const double* src0 = a.feat + (t0 + jlo) * a.ld_feat + a.col0 + db * 64;
  const bool wide = ((a.ld_feat | (int64_t)a.col0 | (int64_t)D) & 1) == 0 &&
                    (reinterpret_cast<uintptr_t>(a.feat) & 15) == 0;
  if (wide) {
    constexpr int CPR = 96;                               // 16-byte chunks per staged row
    constexpr int NLD = (MAXR * CPR + NTHR - 1) / NTHR;
    double2 v[NLD];
    const int total = rows * CPR;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = (int)threadIdx.x + i * NTHR;
      const int row = idx / CPR, rem = idx - row * CPR, w = rem >> 5, c = rem & 31;
      v[i] = make_double2(0.0, 0.0);
      if (idx < total && 2 * c < dblk)
        v[i] = *reinterpret_cast<const double2*>(src0 + (int64_t)row * a.ld_feat + w * D + 2 * c);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = (int)threadIdx.x + i * NTHR;
      const int row = idx / CPR, rem = idx - row * CPR, w = rem >> 5, c = rem & 31;
      if (idx < total) *reinterpret_cast<double2*>(tile + row * ST_W + w * 64 + 2 * c) = v[i];
    }
  } else {
    constexpr int CPR = 192;
    constexpr int NLD = (MAXR * CPR + NTHR - 1) / NTHR;
    const int total = rows * CPR;
    for (int i0 = 0; i0 < NLD; i0 += 8) {
      double v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = (int)threadIdx.x + (i0 + i) * NTHR;
        const int row = idx / CPR, rem = idx - row * CPR, w = rem >> 6, c = rem & 63;
        v[i] = 0.0;
        if (idx < total && c < dblk) v[i] = src0[(int64_t)row * a.ld_feat + w * D + c];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = (int)threadIdx.x + (i0 + i) * NTHR;
        const int row = idx / CPR, rem = idx - row * CPR, w = rem >> 6, c = rem & 63;
        if (idx < total) tile[row * ST_W + w * 64 + c] = v[i];
      }
    }
  }
}

// b of this wave's chunk from the staged rows (the same code path as from memory: fu_form_b with
// the tile's pitch and piece offsets)
template <int FU_FL>
__device__ __forceinline__ void st_form_b_staged(const double* tile, int64_t jlo, const StChunk<FU_FL>& q,
                                                 double (&b)[FU_FL]) {
  MlpgArgs la{};
  la.ld_feat = ST_W;
  la.dim = 64;
  const double* f = tile - jlo * ST_W + (threadIdx.x & 63);
  fu_form_b<FU_FL>(la, f, q.j0, q.n, q.T, q.cst, q.v0, q.v1, q.v2, b);
}

// First launch of the stream path: the shared factor (blocks < nblk) and, beside it, one record per
// group of GA (reduce kernel) and of GB (solve kernel) chunks, expanded from the per-utterance tables
template <int GA, int GB>
__global__ __launch_bounds__(64) void mlpg_prep_kernel(MlpgArgs a, int t_max, int nblk,
                                                       const int* __restrict__ chunk0,
                                                       const int* __restrict__ group_a,
                                                       const int* __restrict__ group_b,
                                                       StRecord* __restrict__ rec_a,
                                                       StRecord* __restrict__ rec_b) {
  if ((int)blockIdx.x < nblk) {        // the shared Cholesky factor of 64 dimensions
    mlpg_factor_block(a, t_max, blockIdx.x);
    return;
  }
  const int u = blockIdx.x - nblk;
  const int64_t t0 = a.offsets[u];
  const int T = (int)(a.offsets[u + 1] - t0);
  const int c0 = chunk0[u];
  const int a0 = group_a[u], na = group_a[u + 1] - a0;
  const int b0 = group_b[u], nb = group_b[u + 1] - b0;
  for (int i = threadIdx.x; i < na + nb; i += 64) {
    const bool second = i >= na;
    const int gi = second ? i - na : i, gw = second ? GB : GA;
    StRecord r{};
    r.t0 = t0;
    r.T = T;
    r.k0 = gi * gw;
    r.chunk = c0 + gi * gw;
    (second ? rec_b : rec_a)[(second ? b0 : a0) + gi] = r;
  }
}

// frames [jlo, jhi) a group of GW chunks starting at chunk k0 needs (one halo row on either side)
template <int FU_FL, int GW>
__device__ __forceinline__ void st_group_rows(const StRecord& rec, int64_t& jlo, int& rows) {
  const int64_t T = rec.T;
  const int K = fu_num_chunks<FU_FL>(T);
  const int kend = rec.k0 + GW < K ? rec.k0 + GW : K;
  jlo = fu_chunk_start<FU_FL>(rec.k0, K, T) - 1;
  if (jlo < 0) jlo = 0;
  int64_t jhi = fu_chunk_start<FU_FL>(kend, K, T) + 1;
  if (jhi > T) jhi = T;
  rows = (int)(jhi - jlo);
}

template <int FU_FL, int GW, bool STAGE>
__global__ __launch_bounds__(GW * 64) void mlpg_reduce_kernel(StreamArgs g) {
  extern __shared__ __attribute__((aligned(16))) double st_tile[];
  const MlpgArgs& a = g.a;
  int64_t jlo = 0;
  StChunk<FU_FL> q;
  const bool active = q.open(g);      // its loads (constants, factor) fly together with the staging loads
  if (STAGE) {
    const int grp = (int)(blockIdx.x / (unsigned)g.nblk), db = (int)(blockIdx.x % (unsigned)g.nblk);
    const StRecord rec = g.rec[grp];
    int rows;
    st_group_rows<FU_FL, GW>(rec, jlo, rows);
    st_stage_rows<GW * 64, GW * FU_FL + 2>(a, db, rec.t0, jlo, rows, st_tile);
    __syncthreads();
  }
  if (!active) return;
  double b[FU_FL];
  if (STAGE) st_form_b_staged<FU_FL>(st_tile, jlo, q, b);
  else fu_form_b<FU_FL>(a, a.feat + q.t0 * a.ld_feat + a.col0 + q.d, q.j0, q.n, q.T, q.cst, q.v0, q.v1, q.v2, b);
  // b goes to the output rows: the solve kernel reads 496 B per frame from there instead of forming
  // b again from 1 488 B of input (and overwrites it with x, chunk by chunk, in place)
  if (q.dok) {
    double* o = a.out + q.t0 * a.ld_out + a.ocol0 + q.d;
#pragma unroll
    for (int i = 0; i < FU_FL; ++i)
      if (q.cst || i < q.n) o[(q.j0 + i) * a.ld_out] = b[i];
  }
  if (q.K == 1) return;               // a one-chunk utterance has nobody to hand a state to
  double e[4], tl[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 0.0};
  FuMats unused;
  if (q.cst) fu_reduce<FU_FL, true, true, false>(q.c, b, q.j0, q.n, tl, e, unused);
  else fu_reduce<FU_FL, false, true, false>(q.c, b, q.j0, q.n, tl, e, unused);
  if (q.dok) {
    const int64_t Dp = (int64_t)g.nblk * 64;
    double* o = g.agg + (int64_t)q.chunk * 4 * Dp + (blockIdx.x % (unsigned)g.nblk) * 64 + (threadIdx.x & 63);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[(int64_t)i * Dp] = e[i];
  }
}

// the data-independent matrices of chunk k of an utterance (any chunk; not inlined: the scan kernel
// calls it from many places and must stay small enough for the instruction cache)
template <int FU_FL>
__device__ __forceinline__ void st_chunk_mats(const FuFac& c, int K, int64_t T, int k, FuMats& m) {
  double none[FU_FL], e4[4], tl[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 0.0};
  const int64_t j0 = fu_chunk_start<FU_FL>(k, K, T), j1 = fu_chunk_start<FU_FL>(k + 1, K, T);
  fu_reduce<FU_FL, false, false, true, true>(c, none, j0, (int)(j1 - j0), tl, e4, m);
}

// The plain sequential scan of one (utterance, 64 dimensions) by one wave: the road the scan
// kernel takes when the factor settles so slowly that the utterance has more non-stationary
// leading chunks than the workgroup has waves to give them.  Correct for anything; not fast.
template <int FU_FL>
__device__ __noinline__ void st_scan_sequential(const FuFac& c, int K, int64_t T, const double* ag, double* st,
                                                int64_t Dp, bool dok) {
  auto lane_cst = [&](int k) {
    const int64_t j0 = fu_chunk_start<FU_FL>(k, K, T), j1 = fu_chunk_start<FU_FL>(k + 1, K, T);
    return (j0 - 2 >= c.ncv) && (j1 <= c.n_shared) && (int)(j1 - j0) == FU_FL;
  };
  FuMats mc;
  {
    double none[FU_FL], e4[4], tl[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 0.0};
    fu_reduce<FU_FL, true, false, true>(c, none, 0, FU_FL, tl, e4, mc);
  }
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < K; ++k) {
    if (dok) { st[((int64_t)k * 4 + 0) * Dp] = s1; st[((int64_t)k * 4 + 1) * Dp] = s2; }
    FuMats m = mc;
    if (!__all(lane_cst(k))) st_chunk_mats<FU_FL>(c, K, T, k, m);
    const double n1 = m.Mf[0] * s1 + m.Mf[1] * s2 + ag[((int64_t)k * 4 + 0) * Dp];
    const double n2 = m.Mf[2] * s1 + m.Mf[3] * s2 + ag[((int64_t)k * 4 + 1) * Dp];
    s1 = n1; s2 = n2;
  }
  double t1 = 0.0, t2 = 0.0;
  for (int k = K - 1; k >= 0; --k) {
    const double si0 = dok ? st[((int64_t)k * 4 + 0) * Dp] : 0.0, si1 = dok ? st[((int64_t)k * 4 + 1) * Dp] : 0.0;
    if (dok) { st[((int64_t)k * 4 + 2) * Dp] = t1; st[((int64_t)k * 4 + 3) * Dp] = t2; }
    FuMats m = mc;
    if (!__all(lane_cst(k))) st_chunk_mats<FU_FL>(c, K, T, k, m);
    const double e0 = ag[((int64_t)k * 4 + 2) * Dp] + m.C[0] * si0 + m.C[1] * si1;
    const double e1 = ag[((int64_t)k * 4 + 3) * Dp] + m.C[2] * si0 + m.C[3] * si1;
    const double n1 = m.Mb[0] * t1 + m.Mb[1] * t2 + e0;
    const double n2 = m.Mb[2] * t1 + m.Mb[3] * t2 + e1;
    t1 = n1; t2 = n2;
  }
}

// One workgroup per (utterance, 64 dimensions): the two affine recurrences over the utterance's
// chunks, as a two-level scan.  The chunks are cut into SW segments in time order, one per wave:
// every non-stationary chunk (the leading ones until all lanes' factors have settled, the last
// two) is a segment of its own, whose wave computes that chunk's matrices; the stationary middle
// is split evenly over the remaining waves, which only ever multiply by the one stationary set.
// A wave folds its segment into (A, q); the SW aggregates meet in LDS; every wave takes the state
// that enters its segment and walks the segment again, now storing.  The chain a wave runs is
// ~K / SW chunks long instead of K, and the aggregates of SB chunks are requested together.
constexpr int ST_SW = 16;      // waves (= segments) per workgroup of the scan kernel

template <int FU_FL>
__global__ __launch_bounds__(ST_SW * 64) void mlpg_scan_kernel(StreamArgs g) {
  constexpr int SW = ST_SW;
  __shared__ double lds_s[SW][6][64];
  const MlpgArgs& a = g.a;
  const int u = (int)(blockIdx.x / (unsigned)g.nblk), db = (int)(blockIdx.x % (unsigned)g.nblk);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D = a.dim;
  const bool dok = db * 64 + lane < D;
  const int d = dok ? db * 64 + lane : D - 1;
  const int64_t T = a.offsets[u + 1] - a.offsets[u];
  if (T <= 0) return;
  const int K = fu_num_chunks<FU_FL>(T);
  const int64_t Dp = (int64_t)g.nblk * 64;
  const int64_t col = (int64_t)db * 64 + lane;
  const int64_t base = g.chunk0[u];
  double* st = g.st + base * 4 * Dp + col;
  const double* ag = g.agg + base * 4 * Dp + col;
  if (K == 1) {
    if (dok && w == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) st[(int64_t)i * Dp] = 0.0;
    }
    return;
  }
  const double v0 = a.var[d], v1 = a.var[D + d], v2 = a.var[2 * D + d];
  FuFac c;
  const int64_t plane = (int64_t)g.t_max * D;
  c.fd = a.scratch + d; c.fl1 = c.fd + plane; c.fl2 = c.fl1 + plane;
  c.ncv = a.nconv[d]; c.n_shared = T >= 3 ? T - 2 : 0; c.T = T; c.D = D;
  c.tau0 = 1.0 / v0; c.tau1_in = 1.0 / v1; c.tau2_in = 1.0 / v2;
  auto lane_cst = [&](int k) {
    const int64_t j0 = fu_chunk_start<FU_FL>(k, K, T), j1 = fu_chunk_start<FU_FL>(k + 1, K, T);
    return (j0 - 2 >= c.ncv) && (j1 <= c.n_shared) && (int)(j1 - j0) == FU_FL;
  };
  // segments: [0, n_lead) one leading chunk each | n_mid waves over [n_lead, tail0) | the last chunks
  int k_settled = 0;                      // first chunk that is stationary for every lane
  while (k_settled < K && !__all(lane_cst(k_settled))) ++k_settled;
  const int tail0 = K - 2 > 0 ? K - 2 : 0;
  const int n_tail = K - tail0;                                    // 1 or 2
  const int n_lead = k_settled < tail0 ? k_settled : tail0;
  if (n_lead > SW - n_tail - 1) {     // see st_scan_sequential
    if (w == 0) st_scan_sequential<FU_FL>(c, K, T, ag, st, Dp, dok);
    return;
  }
  const int n_mid = SW - n_lead - n_tail;
  const int mid_chunks = tail0 - n_lead;
  const int L = (mid_chunks + n_mid - 1) / (n_mid > 0 ? n_mid : 1);
  int k_lo, k_hi;                          // this wave's segment
  const bool single = w < n_lead || w >= n_lead + n_mid;
  if (w < n_lead) { k_lo = w; k_hi = w + 1; }
  else if (w >= n_lead + n_mid) { k_lo = tail0 + (w - n_lead - n_mid); k_hi = k_lo + 1; }
  else {
    const int mw = w - n_lead;
    k_lo = n_lead + mw * L; k_hi = k_lo + L;
    if (k_lo > tail0) k_lo = tail0;
    if (k_hi > tail0) k_hi = tail0;
  }
  FuMats mm;      // single-chunk wave: that chunk's matrices; middle wave: the stationary set
  if (single) {
    st_chunk_mats<FU_FL>(c, K, T, k_lo, mm);
  } else {
    double none[FU_FL], e4[4], tl[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 0.0};
    fu_reduce<FU_FL, true, false, true>(c, none, 0, FU_FL, tl, e4, mm);
  }
  auto mats = [&](int, FuMats& m) { m = mm; };
  constexpr int SB = 8;

  // ---- forward: s_in(k + 1) = M_f(k) s_in(k) + e_f(k)
  double A[4] = {1.0, 0.0, 0.0, 1.0}, q[2] = {0.0, 0.0};
  for (int kb = k_lo; kb < k_hi; kb += SB) {
    double ef[SB][2];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int k = kb + i < k_hi ? kb + i : k_hi - 1;
      ef[i][0] = ag[((int64_t)k * 4 + 0) * Dp];
      ef[i][1] = ag[((int64_t)k * 4 + 1) * Dp];
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      if (kb + i < k_hi) {
        FuMats m;
        mats(kb + i, m);
        const double a0 = m.Mf[0] * A[0] + m.Mf[1] * A[2], a1 = m.Mf[0] * A[1] + m.Mf[1] * A[3];
        const double a2 = m.Mf[2] * A[0] + m.Mf[3] * A[2], a3 = m.Mf[2] * A[1] + m.Mf[3] * A[3];
        const double q0 = m.Mf[0] * q[0] + m.Mf[1] * q[1] + ef[i][0];
        const double q1 = m.Mf[2] * q[0] + m.Mf[3] * q[1] + ef[i][1];
        A[0] = a0; A[1] = a1; A[2] = a2; A[3] = a3; q[0] = q0; q[1] = q1;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) lds_s[w][i][lane] = A[i];
  lds_s[w][4][lane] = q[0];
  lds_s[w][5][lane] = q[1];
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < w; ++i) {
    const double n1 = lds_s[i][0][lane] * s1 + lds_s[i][1][lane] * s2 + lds_s[i][4][lane];
    const double n2 = lds_s[i][2][lane] * s1 + lds_s[i][3][lane] * s2 + lds_s[i][5][lane];
    s1 = n1; s2 = n2;
  }
  for (int kb = k_lo; kb < k_hi; kb += SB) {
    double ef[SB][2];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int k = kb + i < k_hi ? kb + i : k_hi - 1;
      ef[i][0] = ag[((int64_t)k * 4 + 0) * Dp];
      ef[i][1] = ag[((int64_t)k * 4 + 1) * Dp];
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int k = kb + i;
      if (k < k_hi) {
        if (dok) { st[((int64_t)k * 4 + 0) * Dp] = s1; st[((int64_t)k * 4 + 1) * Dp] = s2; }
        FuMats m;
        mats(k, m);
        const double n1 = m.Mf[0] * s1 + m.Mf[1] * s2 + ef[i][0];
        const double n2 = m.Mf[2] * s1 + m.Mf[3] * s2 + ef[i][1];
        s1 = n1; s2 = n2;
      }
    }
  }
  // ---- backward: t_in(k - 1) = M_b(k) t_in(k) + e_b0(k) + C(k) s_in(k)
  // (a lane reads back the s_in it stored above: same thread, same address, program order)
  A[0] = 1.0; A[1] = 0.0; A[2] = 0.0; A[3] = 1.0; q[0] = q[1] = 0.0;
  for (int kb = k_hi - 1; kb >= k_lo; kb -= SB) {
    double eb[SB][2], si[SB][2];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int k = kb - i >= k_lo ? kb - i : k_lo;
      eb[i][0] = ag[((int64_t)k * 4 + 2) * Dp];
      eb[i][1] = ag[((int64_t)k * 4 + 3) * Dp];
      si[i][0] = dok ? st[((int64_t)k * 4 + 0) * Dp] : 0.0;
      si[i][1] = dok ? st[((int64_t)k * 4 + 1) * Dp] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      if (kb - i >= k_lo) {
        FuMats m;
        mats(kb - i, m);
        const double e0 = eb[i][0] + m.C[0] * si[i][0] + m.C[1] * si[i][1];
        const double e1 = eb[i][1] + m.C[2] * si[i][0] + m.C[3] * si[i][1];
        const double a0 = m.Mb[0] * A[0] + m.Mb[1] * A[2], a1 = m.Mb[0] * A[1] + m.Mb[1] * A[3];
        const double a2 = m.Mb[2] * A[0] + m.Mb[3] * A[2], a3 = m.Mb[2] * A[1] + m.Mb[3] * A[3];
        const double q0 = m.Mb[0] * q[0] + m.Mb[1] * q[1] + e0;
        const double q1 = m.Mb[2] * q[0] + m.Mb[3] * q[1] + e1;
        A[0] = a0; A[1] = a1; A[2] = a2; A[3] = a3; q[0] = q0; q[1] = q1;
      }
    }
  }
  __syncthreads();      // every wave has read the forward segment aggregates
#pragma unroll
  for (int i = 0; i < 4; ++i) lds_s[w][i][lane] = A[i];
  lds_s[w][4][lane] = q[0];
  lds_s[w][5][lane] = q[1];
  __syncthreads();
  double t1 = 0.0, t2 = 0.0;
  for (int i = SW - 1; i > w; --i) {
    const double n1 = lds_s[i][0][lane] * t1 + lds_s[i][1][lane] * t2 + lds_s[i][4][lane];
    const double n2 = lds_s[i][2][lane] * t1 + lds_s[i][3][lane] * t2 + lds_s[i][5][lane];
    t1 = n1; t2 = n2;
  }
  for (int kb = k_hi - 1; kb >= k_lo; kb -= SB) {
    double eb[SB][2], si[SB][2];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int k = kb - i >= k_lo ? kb - i : k_lo;
      eb[i][0] = ag[((int64_t)k * 4 + 2) * Dp];
      eb[i][1] = ag[((int64_t)k * 4 + 3) * Dp];
      si[i][0] = dok ? st[((int64_t)k * 4 + 0) * Dp] : 0.0;
      si[i][1] = dok ? st[((int64_t)k * 4 + 1) * Dp] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int k = kb - i;
      if (k >= k_lo) {
        if (dok) { st[((int64_t)k * 4 + 2) * Dp] = t1; st[((int64_t)k * 4 + 3) * Dp] = t2; }
        FuMats m;
        mats(k, m);
        const double e0 = eb[i][0] + m.C[0] * si[i][0] + m.C[1] * si[i][1];
        const double e1 = eb[i][1] + m.C[2] * si[i][0] + m.C[3] * si[i][1];
        const double n1 = m.Mb[0] * t1 + m.Mb[1] * t2 + e0;
        const double n2 = m.Mb[2] * t1 + m.Mb[3] * t2 + e1;
        t1 = n1; t2 = n2;
      }
    }
  }
}

template <int FU_FL, int GW>
__global__ __launch_bounds__(GW * 64) void mlpg_solve_kernel(StreamArgs g) {
  const MlpgArgs& a = g.a;
  StChunk<FU_FL> q;
  if (!q.open(g)) return;
  const int64_t Dp = (int64_t)g.nblk * 64;
  const double* st = g.st + (int64_t)q.chunk * 4 * Dp + (blockIdx.x % (unsigned)g.nblk) * 64 + (threadIdx.x & 63);
  const double s1 = st[0], s2 = st[Dp], t1 = st[2 * Dp], t2 = st[3 * Dp];
  double* o = a.out + q.t0 * a.ld_out + a.ocol0 + q.d;
  double b[FU_FL];      // left in the output rows by the reduce kernel
#pragma unroll
  for (int i = 0; i < FU_FL; ++i) b[i] = (q.cst || i < q.n) ? o[(q.j0 + i) * a.ld_out] : 0.0;
  double M[4], e[2], tl[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 0.0};
  if (q.cst) {
    fu_fwd<FU_FL, true, false>(q.c, b, q.j0, q.n, s1, s2, M, e, tl);
    fu_bwd<FU_FL, true, false>(q.c, b, q.j0, q.n, t1, t2, M, e, tl, o, a.ld_out, q.dok);
  } else {
    fu_fwd<FU_FL, false, false>(q.c, b, q.j0, q.n, s1, s2, M, e, tl);
    fu_bwd<FU_FL, false, false>(q.c, b, q.j0, q.n, t1, t2, M, e, tl, o, a.ld_out, q.dok);
  }
}

}  // namespace itts

using namespace itts;


// reduce -> scan -> solve (see above)
template <int FL, int GW, bool STAGE>
static int mlpg_stream_launch(MlpgArgs a, const int64_t* h_offsets, int n_utts, int dim, int64_t t_max,
                              hipStream_t s) {
  // per utterance: first chunk and first group (batch-wide indices); the per-group records are
  // expanded from them on the device (at 4 096 utterances the host would otherwise build and
  // upload 2.4 - 4.9 MB of records per call)
  constexpr int GS = 4;      // chunks per workgroup of the solve kernel (no LDS there: four waves)
  std::vector<int> tab(3 * (size_t)(n_utts + 1), 0);
  int* chunk0 = tab.data();
  int* group0 = tab.data() + (n_utts + 1);
  int* sgroup0 = tab.data() + 2 * (n_utts + 1);
  int n_chunks = 0, n_groups = 0, n_sgroups = 0;
  for (int u = 0; u < n_utts; ++u) {
    const int64_t T = h_offsets[u + 1] - h_offsets[u];
    chunk0[u] = n_chunks;
    group0[u] = n_groups;
    sgroup0[u] = n_sgroups;
    const int K = T > 0 ? fu_num_chunks<FL>(T) : 0;
    n_chunks += K;
    n_groups += (K + GW - 1) / GW;
    n_sgroups += (K + GS - 1) / GS;
  }
  chunk0[n_utts] = n_chunks;
  group0[n_utts] = n_groups;
  sgroup0[n_utts] = n_sgroups;
  const int nblk = (dim + 63) / 64;
  const size_t rrec_bytes = (size_t)n_groups * sizeof(StRecord);
  const size_t rec_bytes = rrec_bytes + (size_t)n_sgroups * sizeof(StRecord);
  // [records | offsets (int64) | chunk / group tables (int) | aggregates | entry states]; offsets and
  // tables travel in ONE upload
  const size_t off_bytes = ((size_t)(n_utts + 1) * sizeof(int64_t) + 31) / 32 * 32;
  const size_t c0_bytes = off_bytes + (tab.size() * sizeof(int) + 31) / 32 * 32;
  const size_t plane_bytes = (size_t)n_chunks * 4 * nblk * 64 * sizeof(double);
  char* blk = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&blk, rec_bytes + c0_bytes + 2 * plane_bytes, s));
  {
    std::vector<char> host(off_bytes + tab.size() * sizeof(int), 0);
    std::memcpy(host.data(), h_offsets, (size_t)(n_utts + 1) * sizeof(int64_t));
    std::memcpy(host.data() + off_bytes, tab.data(), tab.size() * sizeof(int));
    const int rc = itts::staged_upload(blk + rec_bytes, host.data(), host.size(), s);
    if (rc) return rc;
  }
  a.offsets = reinterpret_cast<const int64_t*>(blk + rec_bytes);
  const int* d_tab = reinterpret_cast<const int*>(blk + rec_bytes + off_bytes);
  hipLaunchKernelGGL((mlpg_prep_kernel<GW, GS>), dim3((unsigned)(nblk + n_utts)), dim3(64), 0, s, a, (int)t_max,
                     nblk, d_tab, d_tab + (n_utts + 1), d_tab + 2 * (n_utts + 1),
                     reinterpret_cast<StRecord*>(blk), reinterpret_cast<StRecord*>(blk + rrec_bytes));
  StreamArgs g;
  g.a = a;
  g.t_max = (int)t_max;
  g.rec = reinterpret_cast<const StRecord*>(blk);
  g.chunk0 = d_tab;
  g.n_groups = n_groups;
  g.nblk = nblk;
  g.agg = reinterpret_cast<double*>(blk + rec_bytes + c0_bytes);
  g.st = g.agg + plane_bytes / sizeof(double);
  const dim3 grid((unsigned)((size_t)n_groups * nblk));
  const size_t tile_bytes = STAGE ? (size_t)(GW * FL + 2) * ST_W * sizeof(double) : 0;
  if (STAGE) {
    static bool attr_set = false;      // more than 64 KB of dynamic LDS needs the attribute once per kernel
    if (!attr_set) {
      ITTS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlpg_reduce_kernel<FL, GW, STAGE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes));
      attr_set = true;
    }
  }
  hipLaunchKernelGGL((mlpg_reduce_kernel<FL, GW, STAGE>), grid, dim3(GW * 64), tile_bytes, s, g);
  hipLaunchKernelGGL(mlpg_scan_kernel<FL>, dim3((unsigned)(n_utts * nblk)), dim3(ST_SW * 64), 0, s, g);
  g.rec = reinterpret_cast<const StRecord*>(blk + rrec_bytes);
  g.n_groups = n_sgroups;
  hipLaunchKernelGGL((mlpg_solve_kernel<FL, GS>), dim3((unsigned)((size_t)n_sgroups * nblk)), dim3(GS * 64), 0, s,
                     g);
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(itts::scratch_free(blk, s));
  return ITTS_OK;
}

extern "C" int64_t itts_mlpg_scratch_bytes(int64_t t_total, int dim) {
  if (t_total < 0 || dim <= 0) return 0;
  // 3 factor planes + device copy of the offsets (<= t_total + 1 entries, padded) + nconv
  return 3 * t_total * (int64_t)dim * 8 + (t_total + 2) * 8 + ((int64_t)dim * 4 + 16) / 8 * 8 + 8;
}

// float32 rows -> the float64 columns col0 .. col0 + 3 dim - 1 in a compact array (for the solves that read doubles)
__global__ void mlpg_widen_kernel(const float* __restrict__ src, int64_t ld, int col0, int cols, int64_t rows, double* __restrict__ dst) {
  const int64_t n = rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    dst[i] = (double)src[r * ld + col0 + (i - r * cols)];
  }
}

// d_feat32 != nullptr: the input rows are float32 (d_feat unused); d_scratch then holds a [Ttot, 3 dim] float64
// array behind the usual scratch, for the batches that do not take the one-pass kernel
// What a call works out from the offsets alone, kept by a caller that solves over the same utterances again (the
// streams of one batch: mcep, lf0, bap; a trainer's fixed validation set): the checks, the longest length and -- for
// the one-pass kernel -- the (start, end) table in launch order, in page-locked memory of its own.  40-50 us of
// host time in front of a 230-us launch otherwise.
struct MlpgPlan {
  std::vector<int64_t> offsets;      // [n_utts + 1]
  int n_utts = 0;
  int64_t t_total = 0, t_max = 0;
  int64_t* table = nullptr;          // hipHostMalloc: (start, end) pairs, longest utterance first
};

static void mlpg_sorted_table(const int64_t* h_offsets, int n_utts, int64_t t_max, int64_t* out) {
  // utterances longest first, equal lengths in their own order: a counting sort over the lengths (a comparison sort
  // of 4 096 utterances was 0.15 ms of host time in front of a 2.8-ms launch)
  std::vector<int> order(n_utts);
  if (t_max <= (int64_t)1 << 20) {
    std::vector<int> start((size_t)t_max + 2, 0);
    for (int u = 0; u < n_utts; ++u) ++start[(size_t)(t_max - (h_offsets[u + 1] - h_offsets[u])) + 1];
    for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
    for (int u = 0; u < n_utts; ++u) order[start[(size_t)(t_max - (h_offsets[u + 1] - h_offsets[u]))]++] = u;
  } else {
    for (int u = 0; u < n_utts; ++u) order[u] = u;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
      return h_offsets[x + 1] - h_offsets[x] > h_offsets[y + 1] - h_offsets[y];
    });
  }
  for (int y = 0; y < n_utts; ++y) {
    out[2 * y] = h_offsets[order[y]];
    out[2 * y + 1] = h_offsets[order[y] + 1];
  }
}

static int mlpg_check_offsets(const int64_t* h_offsets, int n_utts, int64_t* t_max) {
  const int64_t t_total = h_offsets[n_utts];
  ITTS_REQUIRE(h_offsets[0] == 0 && t_total >= 0, "offsets must start at 0");
  for (int u = 0; u < n_utts; ++u)
    ITTS_REQUIRE(h_offsets[u + 1] >= h_offsets[u], "offsets must be non-decreasing");
  ITTS_REQUIRE(n_utts <= t_total + 1, "more utterances than frames");
  *t_max = 0;
  for (int u = 0; u < n_utts; ++u) *t_max = std::max(*t_max, h_offsets[u + 1] - h_offsets[u]);
  return ITTS_OK;
}

static int mlpg_generation_impl(const double* d_feat, const float* d_feat32, int64_t ld_feat, int col0, int dim,
                                const double* d_var, const int64_t* h_offsets, int n_utts,
                                double* d_out, int64_t ld_out, int ocol0, void* d_scratch,
                                void* stream, const MlpgPlan* plan = nullptr) {
  ITTS_REQUIRE(d_var && h_offsets && (n_utts == 0 || ((d_feat || d_feat32) && d_out && d_scratch)), "null pointer");
  ITTS_REQUIRE(dim > 0 && n_utts >= 0 && col0 >= 0 && ocol0 >= 0, "bad sizes");
  ITTS_REQUIRE(ld_feat >= col0 + 3 * (int64_t)dim && ld_out >= ocol0 + (int64_t)dim,
               "leading dimension too small");
  if (n_utts == 0) return ITTS_OK;
  const int64_t t_total = h_offsets[n_utts];
  int64_t t_max = 0;
  if (plan) {
    t_max = plan->t_max;
  } else {
    const int rc = mlpg_check_offsets(h_offsets, n_utts, &t_max);
    if (rc) return rc;
  }
  if (t_total == 0) return ITTS_OK;
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  double* scratch = reinterpret_cast<double*>(d_scratch);
  int64_t* d_off = reinterpret_cast<int64_t*>(scratch + 3 * t_total * (int64_t)dim);
  int* d_nconv = reinterpret_cast<int*>(d_off + (t_total + 2));
  MlpgArgs a{d_feat, ld_feat, col0, dim, d_var, d_off, d_out, ld_out, ocol0, scratch, t_total, d_nconv};
  // reduce -> scan -> solve with 16-frame chunks, two chunks per workgroup, input rows staged through
  // LDS (uploads its own tables and computes the factor in its first launch); batches of short
  // utterances: the sequential sweeps are as fast
  // .. or, from about half a chip's worth of (utterance, 64 dimensions) units, one pass with the right-hand side in LDS
  // (mlpg_ring_kernel: a sequential sweep per unit -- 200 us for a 2 000-frame utterance however few there are --
  // so small batches stay with the form above, which also divides an utterance among workgroups: 16 utterances 132
  // against 225 us, 64: 174 / 201, 256: 372 / 307, 4 096: 4 520 / 3 608).  ITTS_MLPG_STREAM=1 / ITTS_MLPG_RING=1 force one.
  const int nblk = (dim + RING_LANES - 1) / RING_LANES;
  // (the environment is read once per process: five getenv calls were 2-3 us of every call)
  static const char* const force_stream = getenv("ITTS_MLPG_STREAM");
  static const char* const force_ring = getenv("ITTS_MLPG_RING");
  static const bool env_narrow = getenv("ITTS_MLPG_NARROW") != nullptr, env_wide = getenv("ITTS_MLPG_WIDE") != nullptr,
                    env_no_nt = getenv("ITTS_MLPG_NO_NT") != nullptr;
  const bool ring = n_utts <= 65535 &&          // (an utterance per blockIdx.y)
                    (force_ring ? true : (force_stream ? false : (int64_t)n_utts * nblk >= MLPG_RING_FROM));
  if (d_feat32 && !(t_max >= MLPG_SEQ_BELOW && ring)) {
    // the other solves read doubles: widen the three column blocks once, behind the usual scratch
    double* wide = reinterpret_cast<double*>(reinterpret_cast<char*>(d_scratch) + itts_mlpg_scratch_bytes(t_total, dim));
    const int cols = 3 * dim;
    hipLaunchKernelGGL(mlpg_widen_kernel, dim3((unsigned)std::min<int64_t>((t_total * cols + 255) / 256, 8192)), dim3(256), 0, s,
                       d_feat32, ld_feat, col0, cols, t_total, wide);
    ITTS_LAUNCH_CHECK();
    a.feat = wide;
    a.ld_feat = cols;
    a.col0 = 0;
  }
  if (t_max >= MLPG_SEQ_BELOW && !ring) return mlpg_stream_launch<16, 2, true>(a, h_offsets, n_utts, dim, t_max, s);
  if (t_max >= MLPG_SEQ_BELOW) {
    static std::atomic<uint64_t> attr_done{0};
    int dev = 0;
    ITTS_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !((attr_done.load() >> dev) & 1)) {
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mlpg_ring_kernel<double, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES));
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mlpg_ring_kernel<double, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES));
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mlpg_ring_kernel<float, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES));
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mlpg_ring_kernel<double, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES));
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mlpg_ring_kernel<float, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES));
      if (dev < 64) attr_done.fetch_or(uint64_t(1) << dev);
    }
    itts::PinnedTable table;          // (nothing between here and the launch returns early: the slot goes back after it)
    const int64_t* bounds = plan ? plan->table : nullptr;
    if (!bounds) {
      std::vector<int64_t> host(2 * (size_t)n_utts);
      mlpg_sorted_table(h_offsets, n_utts, t_max, host.data());
      const int rc = itts::pinned_table_begin(host.data(), host.size() * sizeof(int64_t), &table);
      if (rc) return rc;
      bounds = static_cast<const int64_t*>(table.p);
    }
    a.offsets = nullptr;          // (the kernel has its bounds in the table)
    RingArgs g{a, bounds, (int)t_max, d_feat32};
    // two dimensions a lane in the helpers (half the memory instructions) where that is what the kernel waits for: float32
    // rows in batches of many rounds of workgroups -- 4 096 utterances 2.68 against 3.23 ms.  With float64 rows the
    // kernel moves 3.8 - 4.1 TB/s either way (3.62 / 3.61 ms), and at 256 utterances the exchange's extra arithmetic
    // costs 3 - 5 % (317 / 301 us; float32 231 / 223).  ITTS_MLPG_WIDE=1 / ITTS_MLPG_NARROW=1 force one (even dim only).
    const bool wide = dim % 2 == 0 && !env_narrow && (env_wide || (d_feat32 && (int64_t)n_utts * nblk >= 1024));
    const dim3 rgrid((unsigned)nblk, (unsigned)n_utts), rblock(RING_THREADS);
    // (float64 rows, y small enough to wait in the memory-side cache: input rows non-temporal -- see RING_LD)
    const bool nt_in = !env_no_nt && (int64_t)t_total * dim * 8 <= (int64_t)192 << 20;
    if (d_feat32 && wide) hipLaunchKernelGGL((mlpg_ring_kernel<float, true, false>), rgrid, rblock, RING_LDS_BYTES, s, g);
    else if (d_feat32) hipLaunchKernelGGL((mlpg_ring_kernel<float, false, false>), rgrid, rblock, RING_LDS_BYTES, s, g);
    else if (wide) hipLaunchKernelGGL((mlpg_ring_kernel<double, true, false>), rgrid, rblock, RING_LDS_BYTES, s, g);
    else if (nt_in) hipLaunchKernelGGL((mlpg_ring_kernel<double, false, true>), rgrid, rblock, RING_LDS_BYTES, s, g);
    else hipLaunchKernelGGL((mlpg_ring_kernel<double, false, false>), rgrid, rblock, RING_LDS_BYTES, s, g);
    const hipError_t launched = hipGetLastError();
    const int rc_table = plan && plan->table ? ITTS_OK : itts::pinned_table_end(&table, s);
    if (launched != hipSuccess) {
      itts::set_error(std::string("mlpg_ring_kernel: ") + hipGetErrorString(launched));
      return ITTS_E_HIP;
    }
    return rc_table;
  }
  {
    const int rc = itts::staged_upload(d_off, h_offsets, (size_t)(n_utts + 1) * sizeof(int64_t), s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(mlpg_factor_kernel, dim3((dim + 63) / 64), dim3(64), 0, s, a, (int)t_max);
  ITTS_LAUNCH_CHECK();
  dim3 grid((dim + MLPG_LANES - 1) / MLPG_LANES, n_utts);
  hipLaunchKernelGGL(mlpg_kernel, grid, dim3(MLPG_LANES), 0, s, a, (int)t_max);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

extern "C" int itts_mlpg_generation(const double* d_feat, int64_t ld_feat, int col0, int dim,
                                    const double* d_var, const int64_t* h_offsets, int n_utts,
                                    double* d_out, int64_t ld_out, int ocol0, void* d_scratch,
                                    void* stream) {
  ITTS_REQUIRE(n_utts == 0 || d_feat, "null pointer");
  return mlpg_generation_impl(d_feat, nullptr, ld_feat, col0, dim, d_var, h_offsets, n_utts, d_out, ld_out, ocol0, d_scratch, stream);
}

extern "C" int64_t itts_mlpg_scratch_bytes_f32(int64_t t_total, int dim) {
  if (t_total < 0 || dim <= 0) return 0;
  return itts_mlpg_scratch_bytes(t_total, dim) + t_total * 3 * (int64_t)dim * 8;
}

extern "C" int itts_mlpg_generation_f32(const float* d_feat, int64_t ld_feat, int col0, int dim,
                                        const double* d_var, const int64_t* h_offsets, int n_utts,
                                        double* d_out, int64_t ld_out, int ocol0, void* d_scratch,
                                        void* stream) {
  ITTS_REQUIRE(n_utts == 0 || d_feat, "null pointer");
  return mlpg_generation_impl(nullptr, d_feat, ld_feat, col0, dim, d_var, h_offsets, n_utts, d_out, ld_out, ocol0, d_scratch, stream);
}

// ---- prepared plans (see MlpgPlan) ----------------------------------------------------------------------------------
extern "C" int itts_mlpg_plan_create(const int64_t* h_offsets, int n_utts, void** plan_out) {
  ITTS_REQUIRE(h_offsets && plan_out && n_utts >= 0, "bad arguments");
  *plan_out = nullptr;
  std::unique_ptr<MlpgPlan> p(new MlpgPlan);
  p->n_utts = n_utts;
  p->offsets.assign(h_offsets, h_offsets + n_utts + 1);
  p->t_total = h_offsets[n_utts];
  if (n_utts > 0) {
    const int rc = mlpg_check_offsets(h_offsets, n_utts, &p->t_max);
    if (rc) return rc;
    ITTS_HIP_CHECK(hipHostMalloc((void**)&p->table, 2 * (size_t)n_utts * sizeof(int64_t), hipHostMallocDefault));
    mlpg_sorted_table(h_offsets, n_utts, p->t_max, p->table);
  }
  *plan_out = p.release();
  return ITTS_OK;
}

extern "C" void itts_mlpg_plan_destroy(void* plan) {
  MlpgPlan* p = static_cast<MlpgPlan*>(plan);
  if (!p) return;
  if (p->table) (void)hipHostFree(p->table);
  delete p;
}

extern "C" int64_t itts_mlpg_plan_frames(const void* plan) {
  return plan ? static_cast<const MlpgPlan*>(plan)->t_total : -1;
}

extern "C" int itts_mlpg_generation_planned(const void* plan, const void* d_feat, int feat_is_f32, int64_t ld_feat,
                                            int col0, int dim, const double* d_var, double* d_out, int64_t ld_out,
                                            int ocol0, void* d_scratch, void* stream) {
  const MlpgPlan* p = static_cast<const MlpgPlan*>(plan);
  ITTS_REQUIRE(p && (p->n_utts == 0 || d_feat), "null pointer");
  // (a launch reads the plan's table in place: the plan must outlive the work queued on `stream`)
  return mlpg_generation_impl(feat_is_f32 ? nullptr : static_cast<const double*>(d_feat),
                              feat_is_f32 ? static_cast<const float*>(d_feat) : nullptr, ld_feat, col0, dim, d_var,
                              p->offsets.data(), p->n_utts, d_out, ld_out, ocol0, d_scratch, stream, p);
}
