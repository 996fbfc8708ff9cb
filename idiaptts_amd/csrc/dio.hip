// DIO F0 estimation (WORLD dio.cpp) for batches of utterances.
// Replaces the first stage of pyworld.wav2world / pyworld.dio
// (src/data_preparation/world/WorldFeatLabelGen.py:792-793, world/LF0LabelGen.py:263-264):
// defaults f0_floor 71, f0_ceil 800, 2 channels per octave, speed 1, allowed_range 0.1.
//
// WORLD filters the whole signal in the frequency domain (a 50 Hz low-cut and one Nuttall
// low-pass per band, each a circular convolution in an FFT long enough to be linear).  On the
// GPU the same linear convolutions are evaluated directly in the time domain with LDS-tiled FIR
// kernels (641 + ~1100 taps per sample at 16 kHz): no 2^18-point FFT, perfectly parallel, and
// slightly more accurate.  The rest follows dio.cpp step by step:
//   dio_events_kernel     negative-going zero crossings of {s, -s, ds, -ds} -> compacted lists
//   dio_candidates_kernel per (band, frame): 4 interpolated interval-f0 -> candidate + score
//   dio_contour_kernel    best band per frame + FixF0Contour's four passes (steps 3/4 are
//                         inherently sequential along the utterance: one lane walks them)
#include <algorithm>
#include <cmath>
#include <vector>

#include "context.h"
#include "world_dev.h"

namespace itts {
using namespace wd;

constexpr double kMaxScore = 100000.0;  // WORLD kMaximumValue
constexpr int MAXB = 16;

struct DioUtt {
  int64_t x_off;     // into x
  int xl;            // samples
  int T;             // frames
  int64_t f_off;     // into f0 output
  int64_t ylc_off;   // into ylc scratch (length yl + 2*pad)
  int64_t sig_off;   // into sig scratch (nb * yl)
  int64_t fine_off;  // into fine scratch (nb*4 lists of cap doubles)
  int cap;           // capacity of one event list
  int64_t cand_off;  // into cands / scores (nb * T each)
  int64_t tmp_off;   // into contour scratch (6*T doubles)
  int64_t cnt_off;   // into counts (nb*4 ints)
};

struct DioParams {
  int fs;
  double frame_period;
  double f0_floor, f0_ceil, allowed_range;
  int nb;
  double bnd[MAXB];
  int hal[MAXB];
  int lowcut_n;   // taps of the 50 Hz low-cut (odd)
  int pad;        // zeros kept on both sides of ylc
  unsigned contour_lds;  // bytes of LDS the contour kernel may use for the candidates
};

// ---- per-utterance mean of y (x zero-extended by one sample) ---------------------------------
__global__ __launch_bounds__(NT) void dio_mean_kernel(const double* __restrict__ x,
                                                      const DioUtt* __restrict__ utts,
                                                      double* __restrict__ mean) {
  __shared__ double red[8];
  const DioUtt u = utts[blockIdx.x];
  const double* xs = x + u.x_off;
  double s = 0.0;
#pragma unroll 8
  for (int i = threadIdx.x; i < u.xl; i += NT) s += xs[i];      // (unrolled: eight loads in flight, same order of additions)
  s = bsum(s, red);
  if (threadIdx.x == 0) mean[blockIdx.x] = s / (double)(u.xl + 1);
}

// ---- FIR: out[o] = sum_k taps[k] * in[o + (L - 1) - k] ----------------------------------------------
// Register-blocked: a thread owns FR consecutive outputs and walks the taps in blocks of FK; per
// block it reads the FR + FK - 1 inputs it needs ONCE from LDS and feeds FR * FK multiply-adds from
// registers, the taps arrive through the scalar cache (uniform address).  One LDS read per multiply-add
// (the first version: every product fetched its sample again) held the kernels at 2.5 % of the
// fp64 rate.  The tile is stored de-interleaved -- element j in plane j % FR at position j / FR --
// so that the lanes of a wave (thread t reads element FR t + const) touch consecutive addresses.
// The tap arrays are padded with zeros to a multiple of FK (a zero tap adds +-0: no value changes)
// and the order of the additions per output is the plain k = 0, 1, 2, ... of the direct sum.
constexpr int FR = 8, FK = 16;
constexpr int TILE = NT * FR;
static_assert(FK % FR == 0, "the plane of a window element must not depend on the tap block");

__host__ __device__ inline int fir_pad_taps(int L) { return (L + FK - 1) / FK * FK; }
// plane length for a tile of TILE outputs and Lp (padded) taps: a multiple of 32 plus 8, which spreads
// the eight planes over the banks when the tile is written
__host__ __device__ inline int fir_plane_len(int Lp) { return ((TILE + Lp - 1 + FR - 1) / FR + 31) / 32 * 32 + 8; }
__host__ __device__ inline size_t fir_lds_bytes(int Lp) { return (size_t)FR * fir_plane_len(Lp) * 8; }

__device__ __forceinline__ void fir_store(double* sy2, int PL, int j, double v) { sy2[(j % FR) * PL + j / FR] = v; }

__device__ __forceinline__ void fir_blocked(const double* sy2, int PL, const double* __restrict__ taps, int Lp,
                                            double (&acc)[FR]) {
  const double* base = sy2 + threadIdx.x;
  for (int k0 = 0; k0 < Lp; k0 += FK) {
    const double* src = base + (Lp - FK - k0) / FR;
    double win[FR + FK - 1];
#pragma unroll
    for (int j = 0; j < FR + FK - 1; ++j) win[j] = src[(j % FR) * PL + j / FR];
#pragma unroll
    for (int kk = 0; kk < FK; ++kk) {
      const double w = taps[k0 + kk];
#pragma unroll
      for (int r = 0; r < FR; ++r) acc[r] += w * win[r + FK - 1 - kk];
    }
  }
}

// low-cut: ylc[n] = y[n] - sum_{k=0}^{N-1} wn[k] * y[n + half - k],  n in [-pad, yl+pad)
__global__ __launch_bounds__(NT) void dio_lowcut_kernel(const double* __restrict__ x,
                                                        const DioUtt* __restrict__ utts,
                                                        const double* __restrict__ mean,
                                                        const double* __restrict__ taps,
                                                        DioParams p, double* __restrict__ ylc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* sy2 = reinterpret_cast<double*>(smem);
  const DioUtt u = utts[blockIdx.y];
  const int yl = u.xl + 1;
  const int total = yl + 2 * p.pad;
  const int o0 = blockIdx.x * TILE;  // index into the padded output
  if (o0 >= total) return;
  const int N = p.lowcut_n, half = (N - 1) / 2, Np = fir_pad_taps(N);
  const int PL = fir_plane_len(Np);
  const double m = mean[blockIdx.y];
  const double* xs = x + u.x_off;
  // input sample index of tile element j: n = (o0 - pad) + j - half - (Np - N)
  const int nbase = o0 - p.pad - half - (Np - N);
  for (int j = threadIdx.x; j < TILE + Np - 1; j += NT) {
    const int n = nbase + j;
    double v = 0.0;
    if (n >= 0 && n < u.xl) v = xs[n] - m;
    else if (n == u.xl) v = -m;
    fir_store(sy2, PL, j, v);
  }
  __syncthreads();
  double acc[FR];
#pragma unroll
  for (int r = 0; r < FR; ++r) acc[r] = 0.0;
  fir_blocked(sy2, PL, taps, Np, acc);
  const int c0 = half + (Np - N);      // tile element of output o's own sample: o + c0
#pragma unroll
  for (int r = 0; r < FR; ++r) {
    const int o = FR * threadIdx.x + r;
    const int jc = o + c0;
    if (o0 + o < total) ylc[u.ylc_off + o0 + o] = sy2[(jc % FR) * PL + jc / FR] - acc[r];
  }
}

// band-pass: sig_b[n] = sum_{k=0}^{4hal-1} lpf_b[k] * ylc[n + 2hal - k], n in [0, yl)
__global__ __launch_bounds__(NT) void dio_band_kernel(const DioUtt* __restrict__ utts,
                                                      const double* __restrict__ lpf_all,
                                                      const int* __restrict__ lpf_off, DioParams p,
                                                      const double* __restrict__ ylc,
                                                      double* __restrict__ sig) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.z;
  const int hal = p.hal[b];
  const int Lp = fir_pad_taps(4 * hal);
  const int PL = fir_plane_len(Lp);
  double* sy2 = reinterpret_cast<double*>(smem);
  const DioUtt u = utts[blockIdx.y];
  const int yl = u.xl + 1;
  const int n0 = blockIdx.x * TILE;
  if (n0 >= yl) return;
  const double* taps = lpf_all + lpf_off[b];
  // tile element j <-> ylc index (n0 + j - (Lp-1) + 2hal) ; stored ylc has `pad` leading zeros
  const double* src = ylc + u.ylc_off + p.pad;
  const int total = yl + p.pad;  // valid indices: [-pad, yl+pad)
  const int base = n0 - (Lp - 1) + 2 * hal;
  for (int j = threadIdx.x; j < TILE + Lp - 1; j += NT) {
    const int n = base + j;
    fir_store(sy2, PL, j, (n >= -p.pad && n < total) ? src[n] : 0.0);
  }
  __syncthreads();
  double acc[FR];
#pragma unroll
  for (int r = 0; r < FR; ++r) acc[r] = 0.0;
  fir_blocked(sy2, PL, taps, Lp, acc);
  double* out = sig + u.sig_off + (int64_t)b * yl;
#pragma unroll
  for (int r = 0; r < FR; ++r) {
    const int n = n0 + FR * threadIdx.x + r;
    if (n < yl) out[n] = acc[r];
  }
}

// ---- zero-crossing events -> compacted "fine" edge positions --------------------------------------
// Four event types per band signal s: v(i) of type 0: s[i]; 1: -s[i]; 2: s[i+1]-s[i] (written as WORLD
// does: (-s[i]) - (-s[i+1])); 3: its negation; an edge of a type sits at i where v(i) > 0 >= v(i+1), and its
// fine position is (i + 1) - v(i) / (v(i+1) - v(i)).  The lists must come out in order, so a list entry's
// slot is the number of edges in front of it.  Rounds 2-3 walked every signal with one workgroup (running
// counts in registers: 49 dependent trips of 2 048 samples, 1.8 ms per analysis at a quarter of the VALU's
// issue rate); now two passes over chunks of ER * ET samples with nothing sequential in them: the
// counts per chunk and type (one read of the 1.4 GB of band signals at HBM speed), a scan of the counts
// per signal, then every chunk again -- its edges compacted and divided by dense lanes.
constexpr int ET = 512;       // threads of an events workgroup
constexpr int ER = 2;         // samples per thread: chunks of 1 024 samples (four per thread: 120 registers, two workgroups per CU)
constexpr int ECH = ER * ET;

// edges of this thread's ER samples of chunk `ch`: a / c = v(i) / v(i+1) of types 0 and 2 (1 and 3: negated),
// the flags, and per wave and sample row the packed counts (16 bits per type)
struct EdgeRow {
  double a[ER][2], c[ER][2];
  bool edge[ER][4];
  int before[ER][4];                 // edges of the type at lower lanes of the wave, same row
  unsigned long long pk[ER];         // the wave's counts of the row, four 16-bit fields
};
__device__ __forceinline__ void edge_rows(const double* __restrict__ s, int yl, int i0, EdgeRow& e) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < ER; ++r) {
    const int i = i0 + r * ET + (int)threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (i < yl - 1) { s0 = s[i]; s1 = s[i + 1]; }
    if (i < yl - 2) s2 = s[i + 2];
    e.a[r][0] = s0; e.c[r][0] = s1;
    e.a[r][1] = (-s0) - (-s1); e.c[r][1] = (-s1) - (-s2);
  }
#pragma unroll
  for (int r = 0; r < ER; ++r) {
    const int i = i0 + r * ET + (int)threadIdx.x;
    unsigned long long pk = 0;
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
      // types 0 / 1 look at i < yl - 1, types 2 / 3 (differences) at i < yl - 2
      const bool in = i < (ty < 2 ? yl - 1 : yl - 2);
      const double av = (ty & 1) ? -e.a[r][ty >> 1] : e.a[r][ty >> 1];
      const double cv = (ty & 1) ? -e.c[r][ty >> 1] : e.c[r][ty >> 1];
      e.edge[r][ty] = in && (av > 0.0) && (cv <= 0.0);
      const unsigned long long bal = __ballot(e.edge[r][ty]);
      e.before[r][ty] = __popcll(bal & ((1ull << lane) - 1ull));
      pk |= (unsigned long long)__popcll(bal) << (16 * ty);
    }
    e.pk[r] = pk;
  }
}

// pass 1: chunk_counts[((u * nb + b) * nch + ch)] = the chunk's edges of the four types (16 bits each)
__global__ __launch_bounds__(ET) void dio_edge_count_kernel(const DioUtt* __restrict__ utts, int nch,
                                                            const double* __restrict__ sig,
                                                            unsigned long long* __restrict__ chunk_counts) {
  constexpr int EW = ET / 64;
  __shared__ unsigned long long wpk[EW];
  const int b = blockIdx.y, nb = gridDim.y;
  const DioUtt u = utts[blockIdx.z];
  const int yl = u.xl + 1;
  const int i0 = blockIdx.x * ECH;
  if (i0 >= yl - 1) return;
  EdgeRow e;
  edge_rows(sig + u.sig_off + (int64_t)b * yl, yl, i0, e);
  unsigned long long tot = 0;
#pragma unroll
  for (int r = 0; r < ER; ++r) tot += e.pk[r];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wpk[wv] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int q = 0; q < EW; ++q) t += wpk[q];
    chunk_counts[((int64_t)blockIdx.z * nb + b) * nch + blockIdx.x] = t;
  }
}

// between the passes: per signal (one wave each), the edges in front of every chunk and the totals
__global__ __launch_bounds__(64) void dio_edge_scan_kernel(const DioUtt* __restrict__ utts, int nb, int nch,
                                                           const unsigned long long* __restrict__ chunk_counts,
                                                           int4* __restrict__ bases, int* __restrict__ counts) {
  const int sgn = blockIdx.x, lane = threadIdx.x;
  const DioUtt u = utts[sgn / nb];
  const int b = sgn % nb;
  const int n = (u.xl + ECH - 1) / ECH;                  // chunks of this signal: i0 < yl - 1 = xl
  unsigned long long clo = 0, chi = 0;                   // running totals: types 0, 1 / 2, 3 as 32-bit fields
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int c = c0 + lane;
    const unsigned long long v = c < n ? chunk_counts[(int64_t)sgn * nch + c] : 0ull;
    const unsigned long long lo = (v & 0xffffull) | (((v >> 16) & 0xffffull) << 32);
    const unsigned long long hi = ((v >> 32) & 0xffffull) | (((v >> 48) & 0xffffull) << 32);
    unsigned long long ilo = lo, ihi = hi;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long a = __shfl_up(ilo, off, 64), d = __shfl_up(ihi, off, 64);
      if (lane >= off) { ilo += a; ihi += d; }
    }
    const unsigned long long elo = clo + ilo - lo, ehi = chi + ihi - hi;
    if (c < n)
      bases[(int64_t)sgn * nch + c] = make_int4((int)(elo & 0xffffffffull), (int)(elo >> 32), (int)(ehi & 0xffffffffull),
                                               (int)(ehi >> 32));
    clo += __shfl(ilo, 63, 64);
    chi += __shfl(ihi, 63, 64);
  }
  if (lane == 0) {
    int* cnt = counts + u.cnt_off + b * 4;
    cnt[0] = (int)(clo & 0xffffffffull); cnt[1] = (int)(clo >> 32);
    cnt[2] = (int)(chi & 0xffffffffull); cnt[3] = (int)(chi >> 32);
  }
}

// pass 2: the fine positions, in order
__global__ __launch_bounds__(ET) void dio_edge_emit_kernel(const DioUtt* __restrict__ utts, int nch,
                                                           const double* __restrict__ sig,
                                                           const int4* __restrict__ bases,
                                                           double* __restrict__ fine) {
  constexpr int EW = ET / 64;
  constexpr int SCAP = ECH / 4;            // edges of one type staged per chunk (more: computed in place)
  __shared__ unsigned long long wpk[ER][EW];
  __shared__ double2 stage_ac[4][SCAP];    // (a, c) of the chunk's edges, compacted, per type
  __shared__ int stage_e[4][SCAP];
  const int b = blockIdx.y, nb = gridDim.y;
  const DioUtt u = utts[blockIdx.z];
  const int yl = u.xl + 1;
  const int i0 = blockIdx.x * ECH;
  if (i0 >= yl - 1) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int4 b4 = bases[((int64_t)blockIdx.z * nb + b) * nch + blockIdx.x];     // in flight with the samples
  EdgeRow e;
  edge_rows(sig + u.sig_off + (int64_t)b * yl, yl, i0, e);
  const int base[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
  for (int r = 0; r < ER; ++r)
    if (lane == 0) wpk[r][wv] = e.pk[r];
  __syncthreads();
  // The interpolation e - a / (c - a) is NOT evaluated where the edge was found (an fp64 division in
  // sixteen branches that nearly every wave enters for a handful of lanes): the chunk's edges are
  // compacted into LDS and divided by dense lanes afterwards.
  unsigned long long run = 0;          // edges of the chunk's rows before r, four 16-bit fields
  int sl[ER][4];
#pragma unroll
  for (int r = 0; r < ER; ++r) {
    const unsigned long long* w = wpk[r];
    unsigned long long off = run, tot = 0;
#pragma unroll
    for (int q = 0; q < EW; ++q) {
      const unsigned long long v = w[q];
      off += q < wv ? v : 0ull;
      tot += v;
    }
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) sl[r][ty] = (int)((off >> (16 * ty)) & 0xffffull) + e.before[r][ty];
    run += tot;
  }
  double* out0 = fine + u.fine_off + (int64_t)(b * 4) * u.cap;
  int trip[4];
#pragma unroll
  for (int ty = 0; ty < 4; ++ty) {
    trip[ty] = (int)((run >> (16 * ty)) & 0xffffull);
#pragma unroll
    for (int r = 0; r < ER; ++r) {
      if (e.edge[r][ty]) {
        const double av = (ty & 1) ? -e.a[r][ty >> 1] : e.a[r][ty >> 1];
        const double cv = (ty & 1) ? -e.c[r][ty >> 1] : e.c[r][ty >> 1];
        const int ei = i0 + r * ET + (int)threadIdx.x + 1;
        const int sp = sl[r][ty];                     // position among this chunk's edges of the type
        if (sp < SCAP) {
          stage_ac[ty][sp] = make_double2(av, cv);
          stage_e[ty][sp] = ei;
        } else if (base[ty] + sp < u.cap) {           // beyond the staging area: in place
          out0[(int64_t)ty * u.cap + base[ty] + sp] = (double)ei - av / (cv - av);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int ty = 0; ty < 4; ++ty) {
    const int nst = trip[ty] < SCAP ? trip[ty] : SCAP;
    if ((int)threadIdx.x < nst && base[ty] + (int)threadIdx.x < u.cap) {
      const double2 ac = stage_ac[ty][threadIdx.x];
      out0[(int64_t)ty * u.cap + base[ty] + threadIdx.x] = (double)stage_e[ty][threadIdx.x] - ac.x / (ac.y - ac.x);
    }
  }
}

// ---- per (band, frame) candidate and score ----------------------------------------------------------
__global__ __launch_bounds__(NT) void dio_candidates_kernel(const DioUtt* __restrict__ utts, DioParams p,
                                                            const double* __restrict__ fine,
                                                            const int* __restrict__ counts,
                                                            double* __restrict__ cands,
                                                            double* __restrict__ scores) {
  const int b = blockIdx.y;
  const DioUtt u = utts[blockIdx.z];
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= u.T) return;
  const double fs = (double)p.fs;
  const double t = (double)i * p.frame_period / 1000.0;
  const int* cnt = counts + u.cnt_off + b * 4;
  bool ok = true;
  for (int k = 0; k < 4; ++k) {
    const int nint = cnt[k] - 1;  // number of intervals
    if (cnt[k] < 2 || nint - 2 <= 0) ok = false;
  }
  double cand = 0.0, score = kMaxScore;
  if (ok) {
    double v[4];
    for (int k = 0; k < 4; ++k) {
      const double* fe = fine + u.fine_off + (int64_t)(b * 4 + k) * u.cap;
      const int n = cnt[k] - 1;  // intervals: loc[j] = (fe[j]+fe[j+1])/2/fs, f0[j] = fs/(fe[j+1]-fe[j])
      // histc: number of loc[j] <= t, clamped to [1, n-1]
      int lo = 0, hi = n;  // find first j with loc[j] > t
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double loc = (fe[mid] + fe[mid + 1]) / 2.0 / fs;
        if (loc <= t) lo = mid + 1; else hi = mid;
      }
      int kk = lo;
      if (kk < 1) kk = 1;
      if (kk > n - 1) kk = n - 1;
      const double x0 = (fe[kk - 1] + fe[kk]) / 2.0 / fs, x1 = (fe[kk] + fe[kk + 1]) / 2.0 / fs;
      const double y0 = fs / (fe[kk] - fe[kk - 1]), y1 = fs / (fe[kk + 1] - fe[kk]);
      const double sfrac = (t - x0) / (x1 - x0);
      v[k] = y0 + sfrac * (y1 - y0);
    }
    const double c = (v[0] + v[1] + v[2] + v[3]) / 4.0;
    double sc = 0.0;
    for (int k = 0; k < 4; ++k) sc += (v[k] - c) * (v[k] - c);
    sc = sqrt(sc / 3.0);
    const double bf = p.bnd[b];
    if (!(c > bf || c < bf / 2.0 || c > p.f0_ceil || c < p.f0_floor)) {
      cand = c;
      score = sc;
    }
  }
  score = score / (cand + kEps);
  cands[u.cand_off + (int64_t)b * u.T + i] = cand;
  scores[u.cand_off + (int64_t)b * u.T + i] = score;
}

// ---- best band + FixF0Contour ------------------------------------------------------------------------
__device__ inline double select_best(double cur, double past, const double* cands, int nb,
                                     int T, int ti, double ar) {
  const double ref = (cur * 3.0 - past) / 2.0;
  double me = fabs(ref - cands[ti]);
  double bf = cands[ti];
  for (int i = 1; i < nb; ++i) {
    const double cv = cands[(int64_t)i * T + ti];
    const double ce = fabs(ref - cv);
    if (ce < me) {
      me = ce;
      bf = cv;
    }
  }
  if (fabs(1.0 - bf / ref) > ar) return 0.0;
  return bf;
}

__global__ __launch_bounds__(NT) void dio_contour_kernel(const DioUtt* __restrict__ utts, DioParams p,
                                                         const double* __restrict__ cands_all,
                                                         const double* __restrict__ scores_all,
                                                         double* __restrict__ tmp_all,
                                                         double* __restrict__ f0_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const DioUtt u = utts[blockIdx.x];
  const int T = u.T, nb = p.nb;
  const double* cands = cands_all + u.cand_off;
  const double* scores = scores_all + u.cand_off;
  double* out = f0_out + u.f_off;
  const double ar = p.allowed_range;
  const int vrm = (int)(0.5 + 1000.0 / p.frame_period / p.f0_floor) * 2 + 1;
  if (T <= vrm) {
    for (int i = threadIdx.x; i < T; i += NT) out[i] = 0.0;
    return;
  }
  // The sequential passes (steps 3 / 4) are chains of dependent reads: keep the candidates and
  // the two working contours in LDS when they fit ((nb + 2) * T doubles), else in global scratch.
  const bool in_lds = (size_t)(nb + 2) * T * 8 <= p.contour_lds;
  double* lds = reinterpret_cast<double*>(smem_c);
  double* base = tmp_all + u.tmp_off;                 // best candidate per frame
  double* s1 = in_lds ? lds + (size_t)nb * T : base + T;          // step-2 contour (boundaries)
  double* s3 = in_lds ? lds + (size_t)(nb + 1) * T : base + 2 * T;  // working contour
  double* tmp = base + 3 * T;                         // step-1 contour
  const double* scand = cands;
  if (in_lds) {
    for (int i = threadIdx.x; i < nb * T; i += NT) lds[i] = cands[i];
    scand = lds;
  }
  // best candidate per frame (first band wins ties), restricted to [vrm, T-vrm)
  for (int i = threadIdx.x; i < T; i += NT) {
    double t = scores[i], bv = cands[i];
    for (int j = 1; j < nb; ++j) {
      const double sj = scores[(int64_t)j * T + i];
      if (t > sj) {
        t = sj;
        bv = cands[(int64_t)j * T + i];
      }
    }
    base[i] = (i >= vrm && i < T - vrm) ? bv : 0.0;
  }
  __syncthreads();
  // step 1: rapid changes
  for (int i = threadIdx.x; i < T; i += NT) {
    double v = 0.0;
    if (i >= vrm) v = fabs((base[i] - base[i - 1]) / (kEps + base[i])) < ar ? base[i] : 0.0;
    tmp[i] = v;
  }
  __syncthreads();
  // step 2: drop frames whose +-c neighbourhood contains an unvoiced frame
  const int c = (vrm - 1) / 2;
  for (int i = threadIdx.x; i < T; i += NT) {
    double v = tmp[i];
    if (i >= c && i < T - c) {
      for (int j = -c; j <= c; ++j)
        if (tmp[i + j] == 0.0) {
          v = 0.0;
          break;
        }
    }
    s1[i] = v;
    s3[i] = v;
  }
  __syncthreads();
  // Voiced->unvoiced ("negative") and unvoiced->voiced ("positive") boundaries of the step-2
  // contour as bit masks, built by all threads: the sequential lane below then walks set bits
  // instead of scanning the contour with dependent loads.  The step-1 buffer is dead by now and
  // holds the masks ([nw] negative, [nw] positive) when they do not fit the static LDS copy.
  const int nw = (T + 63) / 64;
  __shared__ unsigned long long mask_lds[2][128];
  unsigned long long* negm = nw <= 128 ? mask_lds[0] : reinterpret_cast<unsigned long long*>(tmp);
  unsigned long long* posm = nw <= 128 ? mask_lds[1] : negm + nw;
  __syncthreads();                                   // everyone is done reading tmp (step 2)
  for (int base_i = 0; base_i < nw * 64; base_i += NT) {
    const int i = base_i + threadIdx.x;
    const bool in = i >= 1 && i < T;
    const double a0 = in ? s1[i - 1] : 0.0, a1 = in ? s1[i] : 0.0;
    const unsigned long long nm = __ballot(in && a1 == 0.0 && a0 != 0.0);
    const unsigned long long pm = __ballot(in && a0 == 0.0 && a1 != 0.0);
    if ((threadIdx.x & 63) == 0 && i / 64 < nw) {
      negm[i / 64] = nm;
      posm[i / 64] = pm;
    }
  }
  __syncthreads();
  // steps 3 and 4 are sequential along time (each extension reads values the previous one wrote)
  if (threadIdx.x == 0) {
    // step 3: forward extension from every negative boundary up to the frame before the next one
    int prev = -1;
    for (int w = 0; w <= nw; ++w) {
      unsigned long long m = w < nw ? negm[w] : 1ull;          // sentinel: flush the last boundary
      while (m) {
        const int bnd = w < nw ? w * 64 + __ffsll((long long)m) - 1 : T;
        m &= m - 1;
        if (prev >= 0) {
          const int limit = bnd >= T ? T - 1 : bnd - 1;
          for (int j = prev - 1; j < limit; ++j) {
            s3[j + 1] = select_best(s3[j], s3[j - 1], scand, nb, T, j + 1, ar);
            if (s3[j + 1] == 0.0) break;
          }
        }
        prev = bnd;
      }
    }
    // step 4: backward extension from every positive boundary down to the previous one (in place)
    prev = -1;
    for (int w = nw - 1; w >= -1; --w) {
      unsigned long long m = w >= 0 ? posm[w] : 1ull;
      while (m) {
        const int hb = 63 - __clzll((long long)m);
        const int bnd = w >= 0 ? w * 64 + hb : 0;
        m &= ~(1ull << hb);
        if (prev >= 0) {
          const int limit = bnd < 1 ? 1 : bnd;
          for (int j = prev; j > limit; --j) {
            s3[j - 1] = select_best(s3[j], s3[j + 1], scand, nb, T, j - 1, ar);
            if (s3[j - 1] == 0.0) break;
          }
        }
        prev = bnd;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T; i += NT) out[i] = s3[i];
}

}  // namespace itts

using namespace itts;

static int mround_h(double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); }

extern "C" int itts_dio(const double* d_x, const int64_t* h_x_off, const int64_t* h_f_off, int n_utts,
                        int fs, double frame_period_ms, double f0_floor, double f0_ceil,
                        double channels_in_octave, double allowed_range, double* d_f0, void* stream) {
  ITTS_REQUIRE(h_x_off && h_f_off && (n_utts == 0 || (d_x && d_f0)), "null pointer");
  ITTS_REQUIRE(n_utts >= 0 && fs > 0 && frame_period_ms > 0, "bad sizes");
  ITTS_REQUIRE(f0_floor > 0 && f0_ceil > f0_floor && channels_in_octave > 0, "bad f0 range");
  if (n_utts == 0) return ITTS_OK;
  hipStream_t s = as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  DioParams p{};
  p.fs = fs; p.frame_period = frame_period_ms; p.f0_floor = f0_floor; p.f0_ceil = f0_ceil;
  p.allowed_range = allowed_range;
  p.nb = 1 + (int)(std::log(f0_ceil / f0_floor) / std::log(2.0) * channels_in_octave);
  ITTS_REQUIRE(p.nb >= 1 && p.nb <= MAXB, "too many DIO bands");
  int lpf_total = 0;
  std::vector<int> lpf_off(p.nb);
  for (int i = 0; i < p.nb; ++i) {
    p.bnd[i] = f0_floor * std::pow(2.0, (i + 1) / channels_in_octave);
    p.hal[i] = mround_h((double)fs / p.bnd[i] / 2.0);
    ITTS_REQUIRE(p.hal[i] >= 1, "sampling rate too low for the DIO bands");
    lpf_off[i] = lpf_total;
    lpf_total += fir_pad_taps(4 * p.hal[i]);      // zero taps up to a whole tap block
  }
  p.lowcut_n = mround_h((double)fs / 50.0) * 2 + 1;
  p.pad = 2 * p.hal[0] + 2;
  // filter taps (host, tiny)
  std::vector<double> taps(fir_pad_taps(p.lowcut_n), 0.0), lpf(lpf_total, 0.0);
  {
    double wsum = 0.0;
    for (int i = 1; i <= p.lowcut_n; ++i) {
      taps[i - 1] = 0.5 - 0.5 * std::cos(i * 2.0 * M_PI / (p.lowcut_n + 1));
      wsum += taps[i - 1];
    }
    for (int i = 0; i < p.lowcut_n; ++i) taps[i] = taps[i] / wsum;
    for (int b = 0; b < p.nb; ++b) {
      const int n = 4 * p.hal[b];
      for (int i = 0; i < n; ++i) {
        const double t = (double)i / (n - 1.0);
        lpf[lpf_off[b] + i] = 0.355768 - 0.487396 * std::cos(2.0 * M_PI * t) +
                              0.144232 * std::cos(4.0 * M_PI * t) - 0.012604 * std::cos(6.0 * M_PI * t);
      }
    }
  }
  double *d_taps = nullptr, *d_lpf = nullptr;
  int* d_lpf_off = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_taps, taps.size() * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_lpf, lpf.size() * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_lpf_off, lpf_off.size() * 4, s));
  // (through the page-locked staging ring: the host vectors may go at once, and nothing waits for the stream)
  if (int rc = itts::staged_upload(d_taps, taps.data(), taps.size() * 8, s)) return rc;
  if (int rc = itts::staged_upload(d_lpf, lpf.data(), lpf.size() * 8, s)) return rc;
  if (int rc = itts::staged_upload(d_lpf_off, lpf_off.data(), lpf_off.size() * 4, s)) return rc;

  const int64_t budget = (int64_t)24 << 30;  // scratch bytes per sub-batch (one sub-batch for 256 utterances: the per-utterance kernels then fill all CUs)
  int u0 = 0;
  while (u0 < n_utts) {
    std::vector<DioUtt> utts;
    int64_t ylc_n = 0, sig_n = 0, fine_n = 0, cand_n = 0, tmp_n = 0, cnt_n = 0;
    int max_yl = 0, max_T = 0;
    int u1 = u0;
    while (u1 < n_utts) {
      const int64_t xl64 = h_x_off[u1 + 1] - h_x_off[u1];
      ITTS_REQUIRE(xl64 > 0 && xl64 < ((int64_t)1 << 30), "utterance length out of range");
      const int xl = (int)xl64, yl = xl + 1;
      const int T = (int)itts_world_num_frames(xl, fs, frame_period_ms);
      ITTS_REQUIRE(h_f_off[u1 + 1] - h_f_off[u1] == T, "frame offsets do not match the frame count");
      DioUtt d{};
      d.x_off = h_x_off[u1]; d.xl = xl; d.T = T; d.f_off = h_f_off[u1];
      d.ylc_off = ylc_n; d.sig_off = sig_n; d.fine_off = fine_n; d.cap = yl / 2 + 2;
      d.cand_off = cand_n; d.tmp_off = tmp_n; d.cnt_off = cnt_n;
      const int64_t add_ylc = yl + 2 * p.pad, add_sig = (int64_t)p.nb * yl,
                    add_fine = (int64_t)p.nb * 4 * d.cap, add_cand = (int64_t)p.nb * T;
      const int64_t bytes = 8 * (ylc_n + sig_n + fine_n + 2 * cand_n + tmp_n + add_ylc + add_sig +
                                 add_fine + 2 * add_cand + 6 * (int64_t)T);
      if (!utts.empty() && bytes > budget) break;
      ylc_n += add_ylc; sig_n += add_sig; fine_n += add_fine; cand_n += add_cand; tmp_n += 6 * (int64_t)T;
      cnt_n += p.nb * 4;
      max_yl = std::max(max_yl, yl); max_T = std::max(max_T, T);
      utts.push_back(d);
      ++u1;
    }
    const int U = (int)utts.size();
    DioUtt* d_utts = nullptr;
    double *d_mean = nullptr, *d_ylc = nullptr, *d_sig = nullptr, *d_fine = nullptr, *d_cand = nullptr,
           *d_score = nullptr, *d_tmp = nullptr;
    int* d_cnt = nullptr;
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_utts, U * sizeof(DioUtt), s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_mean, U * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_ylc, ylc_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_sig, sig_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_fine, fine_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_cand, cand_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_score, cand_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_tmp, tmp_n * 8, s));
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_cnt, cnt_n * 4, s));
    if (int rc = itts::staged_upload(d_utts, utts.data(), U * sizeof(DioUtt), s)) return rc;

    hipLaunchKernelGGL(dio_mean_kernel, dim3(U), dim3(NT), 0, s, d_x, d_utts, d_mean);
    ITTS_LAUNCH_CHECK();
    {
      const int total = max_yl + 2 * p.pad;
      const size_t lds = fir_lds_bytes(fir_pad_taps(p.lowcut_n));
      ITTS_REQUIRE(lds <= 160 * 1024, "sampling rate too high for the DIO low-cut tile");
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)dio_lowcut_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(dio_lowcut_kernel, dim3((total + TILE - 1) / TILE, U), dim3(NT), lds, s, d_x,
                         d_utts, d_mean, d_taps, p, d_ylc);
      ITTS_LAUNCH_CHECK();
    }
    {
      const size_t lds = fir_lds_bytes(fir_pad_taps(4 * p.hal[0]));
      ITTS_REQUIRE(lds <= 160 * 1024, "sampling rate too high for the DIO band tile");
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)dio_band_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(dio_band_kernel, dim3((max_yl + TILE - 1) / TILE, U, p.nb), dim3(NT), lds, s,
                         d_utts, d_lpf, d_lpf_off, p, d_ylc, d_sig);
      ITTS_LAUNCH_CHECK();
    }
    {
      const int nch = (max_yl - 1 + ECH - 1) / ECH;        // chunks of the longest signal
      unsigned long long* d_cc = nullptr;
      ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_cc, (size_t)U * p.nb * nch * sizeof(unsigned long long), s));
      int4* d_bases = nullptr;
      ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_bases, (size_t)U * p.nb * nch * sizeof(int4), s));
      hipLaunchKernelGGL(dio_edge_count_kernel, dim3(nch, p.nb, U), dim3(ET), 0, s, d_utts, nch, d_sig, d_cc);
      ITTS_LAUNCH_CHECK();
      hipLaunchKernelGGL(dio_edge_scan_kernel, dim3(U * p.nb), dim3(64), 0, s, d_utts, p.nb, nch, d_cc, d_bases, d_cnt);
      ITTS_LAUNCH_CHECK();
      hipLaunchKernelGGL(dio_edge_emit_kernel, dim3(nch, p.nb, U), dim3(ET), 0, s, d_utts, nch, d_sig, d_bases, d_fine);
      ITTS_LAUNCH_CHECK();
      ITTS_HIP_CHECK(itts::scratch_free(d_cc, s));
      ITTS_HIP_CHECK(itts::scratch_free(d_bases, s));
    }
    hipLaunchKernelGGL(dio_candidates_kernel, dim3((max_T + NT - 1) / NT, p.nb, U), dim3(NT), 0, s, d_utts,
                       p, d_fine, d_cnt, d_cand, d_score);
    ITTS_LAUNCH_CHECK();
    {
      size_t lds = std::min<size_t>((size_t)(p.nb + 2) * max_T * 8, 150 * 1024);
      p.contour_lds = (unsigned)lds;
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)dio_contour_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(dio_contour_kernel, dim3(U), dim3(NT), lds, s, d_utts, p, d_cand, d_score, d_tmp,
                         d_f0);
    }
    ITTS_LAUNCH_CHECK();
    ITTS_HIP_CHECK(itts::scratch_free(d_utts, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_mean, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_ylc, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_sig, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_fine, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_cand, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_score, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_tmp, s));
    ITTS_HIP_CHECK(itts::scratch_free(d_cnt, s));
    u0 = u1;
  }
  ITTS_HIP_CHECK(itts::scratch_free(d_taps, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_lpf, s));
  ITTS_HIP_CHECK(itts::scratch_free(d_lpf_off, s));
  return ITTS_OK;
}
