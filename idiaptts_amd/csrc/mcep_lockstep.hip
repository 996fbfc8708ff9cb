// SPTK mcep (pysptk.mcep, AudioProcessing.py:146-152) in LOCKSTEP over all frames of a batch.
//
// The reference's Newton iteration per frame: c' = freqt(mc, -a), FFT, d = x / exp(2 Re C'), inverse FFT,
// cr = frqtr(r, +a), a 60 x 60 Toeplitz-plus-Hankel solve; initial value mc = freqt(ifft(log x), +a).
// Everything but the ratio and the solve is LINEAR, so over a batch of T frames it is three matrix
// products on the fp64 matrix core (v_mfma_f64_16x16x4_f64), the transforms folded into the warping
// matrices (FreqtTables in context.h):
//     mc0 = LG [T x 513] . initT [513 x 60]                      once      (LG = log periodogram)
//     d   = X / exp(2 MC [T x 60] . specT [60 x 513])            per iteration, ratio in the epilogue
//     cr  = D  [T x 513] . crT  [513 x 119]                      per iteration
// with the solve one WAVE per frame (mcls_solve_dpp_kernel).  All frames advance one Newton step per
// round; converged frames are frozen by a flag and dropped from the work list.  History: round 1 ran
// one workgroup per frame (bound by re-reading 735 KB of warping matrices per frame and iteration),
// round 2 made the warping steps products, rounds 2-4 kept two transforms per frame and iteration in a
// kernel of their own (7.7 ms per analysis, 3.4 ms on wave_fft.h) until round 4 folded them away.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <utility>
#include <vector>

#include "context.h"
#include "world_dev.h"

namespace itts {
using namespace wd;

typedef double f64x4 __attribute__((ext_vector_type(4)));

// C[rows, N] = A[rows, K] . B[K, N]; all row-major f64. Workgroup: 128 rows x 64 cols, wave: 32 x 64
// (two 16-row blocks share every B value a lane loads: B is 80 % of a wave's operand bytes), the
// operands of K chunk s + 1 are requested before the products of chunk s (two register sets used
// alternately).  Used for short K and small row counts; long K: gemm_f64_staged_kernel below.
// `rows` (optional) lists the physical row of every logical row: the Newton rounds only touch the
// frames that have not converged yet.  VEC_A: a lane's four consecutive k of a chunk come as two
// 16-byte loads (needs an even lda and a 16-byte aligned base; rows may be read up to 3 doubles
// past K, the callers' buffers have that slack).
// RATIO: the result is not stored as it is but as  aux[row][col] / exp(2 * result)  -- the Newton loop's
// spectral ratio |X|^2 / |H|^2 formed where the log spectrum leaves the matrix core (aux: the
// periodogram rows, same pitch and row list as C), instead of in a pass of its own over 1.3 GB.
// MGC2SP: nothing is stored to C; the value goes to o64, exp(float(value)) to o32 and its square as a
// double to opow (any of the three, rows of pitch N): mgc2sp's outputs (AudioProcessing.py:252-256, :925).
// workgroups of gemm_f64_kernel for T rows and N columns: per XCD, ceil(row blocks / 8) x column tiles
static inline unsigned gemm_f64_grid(int64_t T, int N) {
  const int64_t nbx = (T + 127) / 128;
  return (unsigned)(8 * ((nbx + 7) / 8) * ((N + 63) / 64));
}

struct GemmOut {
  float* o32;
  double* o64;
  double* opow;
};
template <bool VEC_A, bool RATIO = false, bool MGC2SP = false>
__global__ __launch_bounds__(256, MGC2SP ? 3 : 2) void gemm_f64_kernel(const double* __restrict__ A, int64_t lda,
                                                       const double* __restrict__ Bm, int64_t ldb,
                                                       double* __restrict__ C, int64_t ldc, int64_t T,
                                                       int N, int K, const int* __restrict__ rows,
                                                       const double* __restrict__ aux = nullptr,
                                                       GemmOut out = GemmOut{nullptr, nullptr, nullptr}) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  // 1-D grid; consecutive workgroups go to the eight XCDs in turn (blockIdx % 8 fixes the XCD), so the tiles are dealt
  // per XCD, column tile fastest: the workgroups that share a row block of A -- and, with an odd row pitch (N = 513),
  // the 128-byte lines that straddle two column tiles of an output row -- run next to each other behind ONE L2 (dealt
  // by blockIdx alone, the nine tiles of a row block sat behind eight different L2s: nine fetches of the A block from
  // HBM, and every straddling output line written in two parts): mgc2sp's product 0.78 -> 0.70 ms.  (One workgroup
  // walking all nine column tiles of its row block instead: 0.96 ms.)  Grid: gemm_f64_grid().
  const int ny = (N + 63) / 64;
  const int64_t nbx = (T + 127) / 128;
  const unsigned local = blockIdx.x >> 3;
  const int64_t rb = (int64_t)(local / ny) * 8 + (blockIdx.x & 7);
  if (rb >= nbx) return;
  const int64_t r0 = rb * 128 + wv * 32;
  if (r0 >= T) return;
  const int c0 = (int)(local % ny) * 64;
  bool rok[2];
  const double* ap[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t row = r0 + 16 * h + lr;
    rok[h] = row < T;
    const int64_t prow = rok[h] ? (rows ? rows[row] : row) : 0;
    ap[h] = A + prow * lda;
  }
  f64x4 acc[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[h][q] = (f64x4){0.0, 0.0, 0.0, 0.0};
  // MFMA q of a lane group covers the columns c0 + 4 lr + q: a lane's four B values of one k are
  // four consecutive columns (one 32-byte load instead of four 8-byte ones) and its four results
  // of one output row likewise (one 32-byte store).
  const int cb = c0 + 4 * lr;
  const bool cfull = cb + 3 < N;               // all four columns inside the matrix
  struct Ops { double av[2][4]; double bv[4][4]; };
  // K permutation: lane (lr, kg) supplies k = 16 s + 4 kg + j on MFMA j of chunk s
  // The loads deliver RAW values from clamped addresses; the zeros for k >= K, rows >= T and columns >= N
  // are put in where the values are used (a select right behind a load waits for the load there and then:
  // with the selects in here the chunk requested ahead was waited for before the products started).
  auto load = [&](int s, Ops& o) {
    if (VEC_A) {
      const int k0 = s + 4 * kg;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double2* p2 = reinterpret_cast<const double2*>(ap[h] + (k0 < K ? k0 : 0));
        const double2 v0 = p2[0], v1 = p2[1];
        o.av[h][0] = v0.x; o.av[h][1] = v0.y; o.av[h][2] = v1.x; o.av[h][3] = v1.y;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = s + 4 * kg + j;
      const int kc = k < K ? k : 0;
      if (!VEC_A) {
#pragma unroll
        for (int h = 0; h < 2; ++h) o.av[h][j] = ap[h][kc];
      }
      const double* brow = Bm + (int64_t)kc * ldb;
      if (cfull) {
        const f64x4 b4 = *reinterpret_cast<const f64x4*>(brow + cb);   // global: dword alignment suffices
#pragma unroll
        for (int q = 0; q < 4; ++q) o.bv[j][q] = b4[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) o.bv[j][q] = brow[cb + q < N ? cb + q : 0];
      }
    }
  };
  auto multiply = [&](const Ops& o, int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool kok = s + 4 * kg + j < K;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double b = (kok && (cfull || cb + q < N)) ? o.bv[j][q] : 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double a = (kok && rok[h]) ? o.av[h][j] : 0.0;
          acc[h][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[h][q], 0, 0, 0);
        }
      }
    }
  };
  if (MGC2SP) {
    // output bound (K <= 64, a [rows, 513] array out): one operand set instead of two alternating ones keeps the
    // kernel at three workgroups per CU, whose stores and loads cover each other (136 registers; 0.70 -> 0.61 ms for
    // 315 k frames; four per CU at 128 registers spill eight: 0.67 ms; the alternating sets at three: 31 spilled, 0.82 ms)
    Ops oa;
    for (int s = 0; s < K; s += 16) {
      load(s, oa);
      multiply(oa, s);
    }
  } else {
  Ops oa, ob;
  load(0, oa);
  for (int s = 0; s < K; s += 32) {
    if (s + 16 < K) load(s + 16, ob);
    multiply(oa, s);
    if (s + 16 < K) {
      if (s + 32 < K) load(s + 32, oa);
      multiply(ob, s + 16);
    }
  }
  }
  // C/D map (f64): col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t orow = r0 + 16 * h + kg + 4 * r;
      if (orow >= T) continue;
      const int64_t prow_o = rows ? rows[orow] : orow;
      double* crow = C + prow_o * ldc + cb;
      double o4[4] = {acc[h][0][r], acc[h][1][r], acc[h][2][r], acc[h][3][r]};
      if (RATIO) {
        const double* xrow = aux + prow_o * ldc + cb;
        double x4[4];
        if (cfull) {
          const f64x4 xv = *reinterpret_cast<const f64x4*>(xrow);
#pragma unroll
          for (int q = 0; q < 4; ++q) x4[q] = xv[q];
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) x4[q] = xrow[cb + q < N ? q : 0];
        }
        // (the reference's expression, oracle/c/sptk.c:135.  A cheaper form -- x * exp(-2 S) with
        // fastmath.h's exp, half the instructions -- was measured and changes nothing, 0.429 against
        // 0.445 ms: the launch writes 1.3 GB and reads 1.3 GB, it runs at 6 TB/s)
#pragma unroll
        for (int q = 0; q < 4; ++q) o4[q] = x4[q] / exp(2.0 * o4[q]);
      }
      if (MGC2SP) {
        // rows of pitch N (odd: 513): a lane's four values are 8-byte aligned only -- stored as one vector
        // of that alignment (the hardware takes unaligned 16-byte stores).  The launch runs at 1.6 TB/s of
        // output either way (0.81 ms for 315 k frames, against 0.46 ms for the RATIO form with its even
        // pitch): the misaligned row segments are what it pays for
        typedef double f64x4u __attribute__((ext_vector_type(4), aligned(8)));
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const int64_t o = prow_o * N + cb;
        float amp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) amp[q] = expf((float)o4[q]);
        if (cfull) {
          if (out.o64) *reinterpret_cast<f64x4u*>(out.o64 + o) = (f64x4u){o4[0], o4[1], o4[2], o4[3]};
          if (out.o32) *reinterpret_cast<f32x4u*>(out.o32 + o) = (f32x4u){amp[0], amp[1], amp[2], amp[3]};
          if (out.opow)
            *reinterpret_cast<f64x4u*>(out.opow + o) = (f64x4u){(double)amp[0] * (double)amp[0], (double)amp[1] * (double)amp[1],
                                                               (double)amp[2] * (double)amp[2], (double)amp[3] * (double)amp[3]};
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (cb + q < N) {
              if (out.o64) out.o64[o + q] = o4[q];
              if (out.o32) out.o32[o + q] = amp[q];
              if (out.opow) out.opow[o + q] = (double)amp[q] * (double)amp[q];
            }
          }
        }
      } else if (cfull) {
        *reinterpret_cast<f64x4*>(crow) = (f64x4){o4[0], o4[1], o4[2], o4[3]};
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (cb + q < N) crow[q] = o4[q];
      }
    }
}

// The same product for long K (the [T x 513] operands of the mel-cepstrum: 1.3 GB per launch), staged:
// a stage is 64 k.  Every wave requests the A values of the whole NEXT stage (32 rows x 64 k = 16 KB)
// before it multiplies the current one, and B goes through LDS (two 32 KB stage buffers, filled by
// the workgroup) so that the products wait for LDS only (the loads in flight all count on vmcnt,
// which completes in order: a B value fetched just in time would wait for the whole A prefetch).
// Zeros for k >= K / rows >= T / columns >= N are put in where the values are USED: a select right
// behind a load waits for the load there and then.  Same K permutation and order of accumulation as
// above: bit-identical results.  Measured (314 881 rows, K = 513): 0.75 ms at N = 60, 1.2 ms at
// N = 119, i.e. 28 - 36 of the 77.6 TFLOP/s scripts/handoff_lab/mfma_f64_rate.hip reaches with the
// same instruction, and 1.8 - 2.2 TB/s of A: the kernel above (one 16-k chunk per wave in flight) took
// 0.86 / 1.5 ms.  What holds it now is the request pattern of A (a lane fetches 2 x 16 bytes of its
// own row: every 128-byte line is asked for by two instructions, four lanes each); loading whole
// lines and transposing through LDS is the next step.
template <bool VEC_A>
__global__ __launch_bounds__(256) void gemm_f64_staged_kernel(const double* __restrict__ A, int64_t lda,
                                                              const double* __restrict__ Bm, int64_t ldb,
                                                              double* __restrict__ C, int64_t ldc, int64_t T,
                                                              int N, int K, const int* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  double* Bs = reinterpret_cast<double*>(gsm);          // [2][64 k][64 cols]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * 128 + wv * 32;
  const int c0 = blockIdx.y * 64;
  bool rok[2];
  const double* ap[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t row = r0 + 16 * h + lr;
    rok[h] = row < T;
    const int64_t prow = rok[h] ? (rows ? rows[row] : row) : 0;
    ap[h] = A + prow * lda;
  }
  f64x4 acc[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[h][q] = (f64x4){0.0, 0.0, 0.0, 0.0};
  struct AOps { double av[4][2][4]; };      // [chunk of the stage][row block][j]
  // The loads deliver RAW values (clamped addresses); zeros for k >= K / rows >= T / columns >= N are put
  // in where the values are used: a select right behind a load would wait for the load there and then,
  // and nothing would be in flight during the products.
  auto load_a = [&](int ks, AOps& o) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k0 = ks + 16 * c + 4 * kg;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (VEC_A) {
          const double2* p2 = reinterpret_cast<const double2*>(ap[h] + (k0 < K ? k0 : 0));
          const double2 v0 = p2[0], v1 = p2[1];
          o.av[c][h][0] = v0.x; o.av[c][h][1] = v0.y; o.av[c][h][2] = v1.x; o.av[c][h][3] = v1.y;
        } else {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) o.av[c][h][jj] = ap[h][k0 + jj < K ? k0 + jj : 0];
        }
      }
    }
  };
  // B stage [64 k][64 cols]: thread t fetches elements t + 256 i (consecutive lanes: consecutive columns)
  auto load_b = [&](int ks, double (&br)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int k = ks + (idx >> 6), col = c0 + (idx & 63);
      br[i] = Bm[(k < K && col < N) ? (int64_t)k * ldb + col : 0];
    }
  };
  auto store_b = [&](int buf, int ks, const double (&br)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const bool ok = ks + (idx >> 6) < K && c0 + (idx & 63) < N;
      Bs[buf * 4096 + idx] = ok ? br[i] : 0.0;
    }
  };
  auto multiply = [&](const AOps& o, int buf, int ks) {
    const double* bs = Bs + buf * 4096 + 4 * lr;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (ks + 16 * c >= K) break;          // the last stage may hold fewer than four chunks
      f64x4 bv[4];
      double am[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) am[h][jj] = (rok[h] && ks + 16 * c + 4 * kg + jj < K) ? o.av[c][h][jj] : 0.0;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) bv[jj] = *reinterpret_cast<const f64x4*>(bs + (16 * c + 4 * kg + jj) * 64);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[h][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[h][jj], bv[jj][q], acc[h][q], 0, 0, 0);
    }
  };
  AOps oa, ob;
  double br[16];
  load_a(0, oa);
  load_b(0, br);
  store_b(0, 0, br);
  __syncthreads();
  const int nst = (K + 63) / 64;
  for (int st = 0; st < nst; st += 2) {
    // even stage: operands oa, B buffer 0; the next stage goes to ob / buffer 1
    if (st + 1 < nst) { load_a((st + 1) * 64, ob); load_b((st + 1) * 64, br); }
    multiply(oa, 0, st * 64);
    if (st + 1 < nst) store_b(1, (st + 1) * 64, br);
    __syncthreads();
    if (st + 1 >= nst) break;
    if (st + 2 < nst) { load_a((st + 2) * 64, oa); load_b((st + 2) * 64, br); }
    multiply(ob, 1, (st + 1) * 64);
    if (st + 2 < nst) store_b(0, (st + 2) * 64, br);
    __syncthreads();
  }
  const int cb = c0 + 4 * lr;
  const bool cfull = cb + 3 < N;
  // C/D map (f64): col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t orow = r0 + 16 * h + kg + 4 * r;
      if (orow >= T) continue;
      const int64_t prow_o = rows ? rows[orow] : orow;
      double* crow = C + prow_o * ldc + cb;
      if (cfull) {
        *reinterpret_cast<f64x4*>(crow) = (f64x4){acc[h][0][r], acc[h][1][r], acc[h][2][r], acc[h][3][r]};
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (cb + q < N) crow[q] = acc[h][q][r];
      }
    }
}

// Long K with 16-byte aligned rows: A and B both through LDS.  The workgroup fetches the A stage
// [128 rows x 64 k] as WHOLE rows of 512 bytes (32 lanes x 16 bytes per row: every 128-byte line is
// asked for once, by one instruction) into registers while it multiplies the stage before, then --
// behind a barrier -- parks it in LDS (row pitch 66 doubles: the lanes of an MFMA operand read fall
// on different banks) with the zeros for k >= K / rows >= T filled in.  Same K permutation and order
// of accumulation as the kernels above: bit-identical results.  Measured: the long-K launches of an
// analysis 5.1 -> 4.5 ms (0.65 ms at N = 60, 1.05 ms at N = 119 for 314 881 rows: 33 - 40 TFLOP/s).
// Serving A from L2 (a test with 1 024 distinct rows) does not change the time, nor does the bank
// spread of B: a stage takes 7.2 us against 4.1 us of products; where the rest goes is not measured yet.
constexpr int GA_PITCH = 66;
// B stage in LDS: the four rows k = 4 g .. 4 g + 3 that one lane group (kg) of an MFMA chunk reads form a
// block of 4 x 64 doubles + 4 of padding, so that the four lane groups of a read fall on different banks
constexpr int GB_BLOCK = 4 * 64 + 4;
constexpr int GEMM_LDS_BYTES = (128 * GA_PITCH + 16 * GB_BLOCK) * 8;
__global__ __launch_bounds__(256) void gemm_f64_lds_kernel(const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ Bm, int64_t ldb,
                                                           double* __restrict__ C, int64_t ldc, int64_t T,
                                                           int N, int K, const int* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  double* As = reinterpret_cast<double*>(gsm);          // [128 rows][GA_PITCH]
  double* Bs = As + 128 * GA_PITCH;                     // [16 blocks of 4 k][GB_BLOCK]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  // 1-D grid, column tile fastest: the workgroups that share a row block of A run next to each other
  // (its second reader finds it in L2)
  const int ny = (N + 63) / 64;
  const int64_t rw0 = (int64_t)(blockIdx.x / ny) * 128;
  const int c0 = (int)(blockIdx.x % ny) * 64;
  // loader role: rows 8 i + (t >> 5), i = 0 .. 15, 16-byte segment t & 31 of the stage's 64 k
  const int lrow = threadIdx.x >> 5, seg = threadIdx.x & 31;
  const double* arow[16];
  unsigned rmask = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t row = rw0 + 8 * i + lrow;
    const bool ok = row < T;
    rmask |= ok ? (1u << i) : 0u;
    arow[i] = A + (ok ? (rows ? rows[row] : row) : 0) * lda;
  }
  f64x4 acc[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[h][q] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double2 ar[16];
  double br[16];
  auto load_piece = [&](int ks, int g) {
    const int k0 = ks + 2 * seg;
    ar[g] = *reinterpret_cast<const double2*>(arow[g] + (k0 < K ? k0 : 0));
    const int idx = threadIdx.x + 256 * g;
    const int k = ks + (idx >> 6), col = c0 + (idx & 63);
    br[g] = Bm[(k < K && col < N) ? (int64_t)k * ldb + col : 0];
  };
  auto load_stage = [&](int ks) {
#pragma unroll
    for (int g = 0; g < 16; ++g) load_piece(ks, g);
  };
  auto park_stage = [&](int ks) {
    const int k0 = ks + 2 * seg;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool rv = (rmask >> i) & 1u;
      double2 v;
      v.x = (rv && k0 < K) ? ar[i].x : 0.0;
      v.y = (rv && k0 + 1 < K) ? ar[i].y : 0.0;
      *reinterpret_cast<double2*>(As + (8 * i + lrow) * GA_PITCH + 2 * seg) = v;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const bool ok = ks + (idx >> 6) < K && c0 + (idx & 63) < N;
      Bs[(idx >> 8) * GB_BLOCK + (idx & 255)] = ok ? br[i] : 0.0;
    }
  };
  // Chunk by chunk with a branch each; pf: the 32 requests of the NEXT stage go out in four groups, one
  // behind each chunk's products (issued in one go in front of the products they took 1.2 - 2.5 us per
  // stage -- in-kernel stamps -- during which the matrix unit had nothing to do).  Two finer forms --
  // the four chunks as straight-line code, and two requests behind every eight products pinned by
  // sched_barrier -- produced NaNs in a few frames with this compiler (not understood) and are not used.
  auto multiply = [&](int ks, bool pf) {
    const double* bs = Bs + kg * GB_BLOCK + 4 * lr;
    const double* as = As + (32 * wv + lr) * GA_PITCH + 4 * kg;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (ks + 16 * c >= K) break;          // the last stage may hold fewer than four chunks
      f64x4 bv[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) bv[jj] = *reinterpret_cast<const f64x4*>(bs + 4 * c * GB_BLOCK + jj * 64);
      double am[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double2 v0 = *reinterpret_cast<const double2*>(as + 16 * h * GA_PITCH + 16 * c);
        const double2 v1 = *reinterpret_cast<const double2*>(as + 16 * h * GA_PITCH + 16 * c + 2);
        am[h][0] = v0.x; am[h][1] = v0.y; am[h][2] = v1.x; am[h][3] = v1.y;
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[h][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[h][jj], bv[jj][q], acc[h][q], 0, 0, 0);
      if (pf) {
#pragma unroll
        for (int g = 0; g < 4; ++g) load_piece(ks + 64, 4 * c + g);
      }
    }
  };
  load_stage(0);
  park_stage(0);
  __syncthreads();
  const int nst = (K + 63) / 64;
  for (int st = 0; st < nst; ++st) {
    multiply(st * 64, st + 1 < nst);
    if (st + 1 >= nst) break;
    __syncthreads();                // everybody has read this stage (and the loads above have landed)
    park_stage((st + 1) * 64);
    __syncthreads();
  }
  const int64_t r0 = rw0 + wv * 32;
  const int cb = c0 + 4 * lr;
  const bool cfull = cb + 3 < N;
  // C/D map (f64): col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t orow = r0 + 16 * h + kg + 4 * r;
      if (orow >= T) continue;
      const int64_t prow_o = rows ? rows[orow] : orow;
      double* crow = C + prow_o * ldc + cb;
      if (cfull) {
        *reinterpret_cast<f64x4*>(crow) = (f64x4){acc[h][0][r], acc[h][1][r], acc[h][2][r], acc[h][3][r]};
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (cb + q < N) crow[q] = acc[h][q][r];
      }
    }
}

// One Newton round's TWO products in one kernel (round 5):  cr = (X / exp(2 MC . specT)) . crT.
// The [rows x K] ratio D never exists in memory.  Until round 4 the first product (gemm_f64_kernel<.., RATIO>)
// wrote it and the second (gemm_f64_lds_kernel) read it back: 2 x 4.1 KB per frame and round on top of the 4.1 KB
// of the periodogram, 26 GB of the analysis' 44 GB, the first launch bound by HBM (6 TB/s) and the second at two
// thirds of the fp64 matrix rate: 7.35 ms per analysis of 256 utterances.  Forms tried on the way (DESIGN.md
// section 13b): a wave of 32 frames with the ratio tile parked in its own LDS and the table operands straight from
// L2 (8.9 ms: every operand an L2 round trip in front of its products; with the periodogram by DMA, padded tables
// and explicit operand prefetch 5.9 ms, but 256 registers and a quarter of the accumulators in scratch); then the
// form below, 4.9 ms.
typedef __attribute__((address_space(3))) char* mcls_lds_p;
__device__ __forceinline__ void mcls_dma_piece(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t dst) {
  // lane l fetches 16 bytes at rsrc.base + voff + soff; the wave's 64 pieces land at LDS byte address dst + 16 l
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(rsrc), "s"(soff), "s"(dst)
               : "memory");
}

// The OPERAND TABLES go through LDS, the ratio stays in registers.
// One workgroup of eight waves per CU; a wave owns 16 frames.  Per chunk of 64 spectral bins the workgroup holds the
// chunk's slice of specP [64 k x 64 bins, 32 KB] and of crP [64 bins x 128, 64 KB] in LDS -- brought by DMA (no
// registers, 1-KB pieces; the slice of the NEXT chunk is requested as soon as the last wave has
// read this one's) and read by all twelve waves, so the L2 sees each table once per workgroup and chunk instead of once
// per wave (all eight waves read the same slice), and an operand is an LDS read away from its product instead of an L2 round trip.
//   P1  the log spectrum TRANSPOSED, S^T[bins x frames] = specP^T . MC^T: the lane that ends up with
//       S[frame j][bins 16 t + 4 kg + r] is the lane that must supply D[frame j][k = 16 t + 4 kg + r] as the A operand
//       of P2 -- the rows of the A operand of P1 are handed the bins in that order (perm below) -- so the ratio
//       D = X / exp(2 S) is formed in registers and stays there: no tile in LDS, no exchange.
//   P2  cr[frames x N2] += D . crP, B operand from LDS.
// Same k of the same lane group on the same product, same order of accumulation as in the two-launch form.

constexpr int F3_SPEC_BYTES = 64 * 64 * 8, F3_CR_BYTES = 64 * 128 * 8, F3_LDS_BYTES = F3_SPEC_BYTES + F3_CR_BYTES;
constexpr int F3_WAVES = 8;          // (twelve -- three per SIMD, 168 registers -- spill the accumulators: 6.1 ms)
template <int NTILES>   // 16-column tiles of cr: 8 (N2 <= 128) or 4 (N2 <= 64)
__global__ __launch_bounds__(64 * F3_WAVES, F3_WAVES / 4) void mcls_fused3_kernel(
    const double* __restrict__ mc, int m1, const double* __restrict__ specP, const double* __restrict__ xp, int64_t ldk,
    const double* __restrict__ crP, double* __restrict__ cr, int64_t nr, int K, int kpad, int N2,
    const int* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(1024))) char fsm[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, kg = lane >> 4;
  const uint32_t l0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(mcls_lds_p)fsm);
  const double* specL = reinterpret_cast<const double*>(fsm);
  const double* crL = reinterpret_cast<const double*>(fsm + F3_SPEC_BYTES);
  const bool spec_wave = wv < 4;       // waves 0 .. 3 fetch the specP slices (8 pieces each), the others the crP slices
  constexpr int CR_PIECES = 64 / (F3_WAVES - 4);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(specP), 0, 64 * kpad * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(crP), 0, kpad * 1024, 0x00020000);
  // specP piece p (0 .. 31): rows k = 2 p, 2 p + 1 of the slice, lane l brings 16 bytes (l & 31) of row 2 p + (l >> 5)
  // crP piece q (0 .. 63): row (bin) q of the slice, lane l brings its 16 bytes l
  auto request_tables = [&](int ch) {
    if (spec_wave) {
      const uint32_t voff = (uint32_t)(lane >> 5) * (uint32_t)kpad * 8u + 16u * (uint32_t)(lane & 31);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = 8 * wv + i;
        mcls_dma_piece(srs, voff, (uint32_t)(2 * p) * (uint32_t)kpad * 8u + (uint32_t)ch * 512u, l0 + 1024u * p);
      }
    } else {
#pragma unroll
      for (int i = 0; i < CR_PIECES; ++i) {
        const int q = CR_PIECES * (wv - 4) + i;
        mcls_dma_piece(crs, 16u * (uint32_t)lane, ((uint32_t)ch * 64u + q) * 1024u, l0 + F3_SPEC_BYTES + 1024u * q);
      }
    }
  };
  const int64_t f0 = (int64_t)blockIdx.x * (16 * F3_WAVES) + wv * 16;
  auto phys = [&](int64_t f) -> int64_t { return rows[f < nr ? f : nr - 1]; };   // (never NULL; frames past the end repeat the last)
  const double* mrow = mc + phys(f0 + j) * m1 + 4 * kg;
  const double* xrow = xp + phys(f0 + j) * ldk + 4 * kg;
  f64x4 acc2[NTILES];
#pragma unroll
  for (int nt = 0; nt < NTILES; ++nt) acc2[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
  // periodogram values of a chunk: bins c0 + 16 t + 4 kg .. + 3 of this lane's frame
  f64x4 xv[4];
  auto request_x = [&](int ch) {
#pragma unroll
    for (int t = 0; t < 4; ++t) xv[t] = *reinterpret_cast<const f64x4*>(xrow + ch * 64 + 16 * t);
  };
  request_tables(0);       // (a spec wave holds specP pieces, the others crP pieces: each waits for its own below)
  request_x(0);
  const int nchunk = kpad / 64;
  const int perm = 4 * (j & 3) + (j >> 2);          // row j of the A operand of P1 stands for bin 16 t + perm
  if (spec_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the first specP slice have landed
  __syncthreads();                                                    // ... and everybody else's
  // Two barriers per chunk.  A wave's products wait for nothing but LDS; what a wave has to WAIT for from memory was
  // requested at least a product phase earlier: the specP slice of chunk c + 1 during the second phase of chunk c,
  // the crP slice of chunk c + 1 during the first phase of chunk c + 1, the periodogram values of chunk c + 1 tile by
  // tile as the ratio of chunk c has consumed them.
  for (int ch = 0; ch < nchunk; ++ch) {
    const int c0 = ch * 64;
    // ---- P1 (transposed)
    f64x4 d[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) d[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
    {
      const double* sa = specL + (4 * kg) * 64 + perm;
      f64x4 mv[2];
      mv[0] = *reinterpret_cast<const f64x4*>(mrow);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) mv[(s + 1) & 1] = *reinterpret_cast<const f64x4*>(mrow + 16 * (s + 1));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            d[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[(16 * s + jj) * 64 + 16 * t], mv[s & 1][jj], d[t], 0, 0, 0);
      }
    }
    if (!spec_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the crP slice have landed
    __syncthreads();          // everybody has read the specP slice (the next one may come) and the crP slice is whole
    if (spec_wave && ch + 1 < nchunk) request_tables(ch + 1);
    // ---- the ratio, in registers, tile by tile: d[t][r] = S[frame j][bin c0 + 16 t + 4 kg + r] -> D (the reference's
    // expression, oracle/c/sptk.c:135) -- and P2 behind it: group tcl, step r: k = c0 + 16 tcl + 4 kg + r, the bin
    // d[tcl][r] belongs to.  The ratio of tile t + 1 (some 300 fp64 instructions) is written in front of the 32 products
    // of group t: the two do not depend on each other and issue side by side.
    const bool more = ch + 1 < nchunk;
    const double* xnext = xrow + (more ? ch + 1 : ch) * 64;          // (the last chunk re-reads itself: no branch in the blocks below)
    auto ratio = [&](int t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) d[t][r] = xv[t][r] / exp(2.0 * d[t][r]);
      xv[t] = *reinterpret_cast<const f64x4*>(xnext + 16 * t);       // consumed after the next chunk's P1
    };
    {
      const double* cb = crL + (4 * kg) * 128 + j;
      const int ngroups = min(4, (K - c0 + 15) / 16);        // the last chunk may hold fewer than four groups of 16 k
      ratio(0);
#pragma unroll
      for (int tcl = 0; tcl < 4; ++tcl) {
        if (tcl < ngroups) {
          // one basic block: the ratio of the next tile and the 4 NTILES products of this group do not depend on
          // each other; the scheduler is told to take them in turn (left alone it puts the ratio in front)
          if (tcl + 1 < 4) ratio(tcl + 1);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < NTILES; ++nt)
              acc2[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[tcl][r], cb[(16 * tcl + r) * 128 + 16 * nt], acc2[nt], 0, 0, 0);
          if (tcl + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 4 * NTILES; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
              __builtin_amdgcn_sched_group_barrier(0x002, 128 / (4 * NTILES) + 1, 0);   // VALU of the ratio
            }
          }
        }
      }
    }
    if (spec_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of the next specP slice have landed
    __syncthreads();          // everybody has read the crP slice, and the next specP slice is whole
    if (!spec_wave && more) request_tables(ch + 1);
  }
  // acc2[nt][r]: frame kg + 4 r of the wave, column 16 nt + j
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t f = f0 + kg + 4 * r;
    if (f >= nr) continue;
    double* crow = cr + (int64_t)rows[f] * N2 + j;
#pragma unroll
    for (int nt = 0; nt < NTILES; ++nt)
      if (16 * nt + j < N2) crow[16 * nt] = acc2[nt][r];
  }
}

__global__ void mcls_iota_kernel(int* __restrict__ rows, int64_t T) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < T; i += (int64_t)gridDim.x * blockDim.x) rows[i] = (int)i;
}

// list of the frames that are still iterating (order irrelevant: frames are independent)
__global__ void mcls_compact_kernel(const int* __restrict__ done, int64_t T, int* __restrict__ rows,
                                    int* __restrict__ count) {
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < T; g += (int64_t)gridDim.x * blockDim.x)
    if (!done[g]) rows[atomicAdd(count, 1)] = (int)g;
}

struct LsArgs {
  const double* in;      // amplitude [T,K] (mode 0) or power envelope [T,K] (mode 1)
  int in_is_power;
  int64_t T;
  int flng, logflng, m;
  double alpha, eps;
  int itr1, itr2;
  double dd;
  double* xp;            // [T, f2+1] periodogram
  double* cbuf;          // [T, f2+1] cepstrum / c' / r (reused)
  double* mc;            // [T, m+1]
  double* cr;            // [T, 2m+1]
  double* sprev;         // [T]
  int* done;             // [T]
  int* iters;            // [T]
  int* n_active;         // [MCLS_COUNTERS x 32]: frames done, counted on MCLS_COUNTERS words a cache line apart (mcls_count_done)
  int iter;              // current Newton iteration (1-based)
  int64_t ldk;           // row pitch of xp / cbuf (K rounded up to even: 16-byte aligned rows)
  const int* rows;       // frames still iterating (NULL: all T)
  int64_t n_rows;
  const double* apow;    // [m + 1] (-alpha)^r: the right-hand side's constant part, tabulated once per call
};

// A frame that stops iterating is counted.  On ONE word the quarter of a million frames that converge in the loop's
// third round queued up in the L2 (that round's solve: 2.83 ms where the rounds around it, with as many frames, take
// 1.95); on 64 words a cache line apart, picked by the workgroup, they do not.  mcls_remaining_kernel sums them for the host.
constexpr int MCLS_COUNTERS = 64;
__device__ __forceinline__ void mcls_count_done(int* counters) {
  atomicAdd(counters + 32 * (blockIdx.x & (MCLS_COUNTERS - 1)), 1);
}
__global__ __launch_bounds__(64) void mcls_remaining_kernel(const int* __restrict__ counters, int T, int* __restrict__ out) {
  int v = counters[32 * threadIdx.x];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if (threadIdx.x == 0) out[0] = T - v;
}



// Periodogram and its logarithm of every frame, one wave per frame: xp = |X|^2 + eps (kept for the loop),
// lg = log xp into cbuf (the initial mel-cepstrum is lg x initT: FreqtTables), and the loop's first
// reference value c[0] / 2 = mean log periodogram / 2 (c = irfft(lg): c[0] = (lg[0] + lg[f2] + 2 sum lg[k]) / flng)
__global__ __launch_bounds__(256) void mcls_init_flat_kernel(LsArgs a) {
  const int lane = threadIdx.x & 63;
  const int f2 = a.flng / 2, K = f2 + 1;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); g < a.T; g += nw) {
    const double* in = a.in + g * K;
    double* xp = a.xp + g * a.ldk;
    double* lg = a.cbuf + g * a.ldk;
    double acc = 0.0;
    for (int k = lane; k < K; k += 64) {
      double v = in[k];
      if (a.in_is_power) v = sqrt(v);  // amp_sp = sqrt(pow_sp), WorldFeatLabelGen.py:795
      const double x = v * v + a.eps;
      const double l = wd::log_pos(x);
      xp[k] = x;
      lg[k] = l;
      acc += (k == 0 || k == f2) ? l : 2.0 * l;
    }
    // the pad columns of the row (even pitch): the fused kernel's tiles cover them (they meet zero rows of crP,
    // which only an Inf or NaN would survive)
    for (int k = K + lane; k < a.ldk; k += 64) xp[k] = 1.0;
    acc = wave_sum(acc);
    if (lane == 0) {
      a.sprev[g] = acc / (double)a.flng / 2;
      a.done[g] = 0;
      a.iters[g] = 0;
    }
  }
}


// The start of the loop in ONE kernel (round 5; until then mcls_init_flat_kernel wrote the periodogram AND its
// logarithm -- 2 x 1.3 GB for 256 utterances -- and a product read the logarithms back): a wave owns 16 frames and
// walks the bins in chunks of 64 as mcls_fused3_kernel does; the logarithms of a chunk are formed in the registers of
// the lanes that supply them as the A operand of  mc0 += LG . initP  (the chunk's slice of initP, 32 KB, in LDS by
// DMA, two buffers), the periodogram is written for the loop, the loop's first reference value (the mean log
// periodogram / 2) is summed on the way.  Same k of the same lane group on the same product, same order of
// accumulation as the product it replaces: the same initial mel-cepstrum bit for bit.  The reference value's sum
// runs in another order (it is only compared against when miniter < 2; pysptk's default is 2).
constexpr int FI_WAVES = 8, FI_SLICE_BYTES = 64 * 64 * 8, FI_LDS_BYTES = 2 * FI_SLICE_BYTES;
__global__ __launch_bounds__(64 * FI_WAVES, 4) void mcls_init_fused_kernel(LsArgs a, const double* __restrict__ initP, int kpad) {
  extern __shared__ __attribute__((aligned(1024))) char fsm[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, kg = lane >> 4;
  const uint32_t l0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(mcls_lds_p)fsm);
  const int f2 = a.flng / 2, K = f2 + 1, m1 = a.m + 1;
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(initP), 0, kpad * 512, 0x00020000);
  // slice of chunk ch: rows (bins) 64 ch .. + 63 of initP, 512 bytes each: piece p = rows 2 p, 2 p + 1; waves 0 .. 3 fetch
  auto request_slice = [&](int ch) {
    if (wv < 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = 8 * wv + i;
        mcls_dma_piece(irs, 16u * (uint32_t)lane, ((uint32_t)ch * 64u + 2u * p) * 512u, l0 + (uint32_t)(ch & 1) * FI_SLICE_BYTES + 1024u * p);
      }
    }
  };
  const int64_t f0 = (int64_t)blockIdx.x * (16 * FI_WAVES) + wv * 16;
  const int64_t fj = f0 + j < a.T ? f0 + j : a.T - 1;              // (frames past the end repeat the last and are not stored)
  const bool fvalid = f0 + j < a.T;
  const double* irow = a.in + fj * K + 4 * kg;
  double* xrow = a.xp + fj * a.ldk + 4 * kg;
  f64x4 acc2[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc2[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double lsum = 0.0;
  const int nchunk = kpad / 64;
  request_slice(0);
  for (int ch = 0; ch < nchunk; ++ch) {
    const int c0 = ch * 64;
    // this chunk's amplitudes -> periodogram -> logarithms, in the layout of the A operand (lane (j, kg): bins
    // c0 + 16 t + 4 kg + r of frame j)
    f64x4 lg[4];
    const bool whole = c0 + 64 <= K;             // (uniform) all 64 bins inside the spectrum
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      double v[4];
      if (whole) {
        typedef double f64x4u __attribute__((ext_vector_type(4), aligned(8)));      // rows of K = 513 doubles: 8-byte aligned
        const f64x4u q = *reinterpret_cast<const f64x4u*>(irow + c0 + 16 * t);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = q[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = c0 + 16 * t + 4 * kg + r < K ? irow[c0 + 16 * t + r] : 1.0;
      }
      double x[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = c0 + 16 * t + 4 * kg + r;
        if (a.in_is_power) v[r] = sqrt(v[r]);    // amp_sp = sqrt(pow_sp), WorldFeatLabelGen.py:795
        x[r] = v[r] * v[r] + a.eps;
        const double l = wd::log_pos(x[r]);
        const bool in = k < K;
        lg[t][r] = in ? l : 0.0;
        if (!in) x[r] = 1.0;                     // pad columns of the row: finite (they meet zero rows of crP in the loop)
        lsum += in ? ((k == 0 || k == f2) ? l : 2.0 * l) : 0.0;
      }
      if (fvalid) {
        if (whole) {
          *reinterpret_cast<double2*>(xrow + c0 + 16 * t) = make_double2(x[0], x[1]);
          *reinterpret_cast<double2*>(xrow + c0 + 16 * t + 2) = make_double2(x[2], x[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (c0 + 16 * t + 4 * kg + r < a.ldk) xrow[c0 + 16 * t + r] = x[r];
        }
      }
    }
    if (wv < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the slice have landed
    __syncthreads();                                                  // ... everybody's; and everybody is done with the other buffer
    if (ch + 1 < nchunk) request_slice(ch + 1);
    const double* ib = reinterpret_cast<const double*>(fsm + (ch & 1) * FI_SLICE_BYTES) + (4 * kg) * 64 + j;
    const int ngroups = min(4, (K - c0 + 15) / 16);
#pragma unroll
    for (int tcl = 0; tcl < 4; ++tcl) {
      if (tcl < ngroups) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc2[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(lg[tcl][r], ib[(16 * tcl + r) * 64 + 16 * nt], acc2[nt], 0, 0, 0);
      }
    }
  }
  // the frame's sum: its bins are spread over the four lanes (j, kg = 0 .. 3)
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (fvalid && kg == 0) {
    a.sprev[fj] = lsum / (double)a.flng / 2;
    a.done[fj] = 0;
    a.iters[fj] = 0;
  }
  // acc2[nt][r]: frame kg + 4 r of the wave, column 16 nt + j
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t f = f0 + kg + 4 * r;
    if (f >= a.T) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      if (16 * nt + j < m1) a.mc[f * m1 + 16 * nt + j] = acc2[nt][r];
  }
}

// convergence test + Newton update of one frame from cr
__global__ __launch_bounds__(NT) void mcls_solve_kernel(LsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t g = a.rows ? a.rows[blockIdx.x] : blockIdx.x;
  if (a.done[g]) return;
  const int m = a.m, m1 = m + 1, m2 = 2 * m, ld = m + 2;
  double* cr = reinterpret_cast<double*>(smem);   // [2m+1]
  double* A = cr + (m2 + 2);                      // [m1][m+2]
  double* fcol = A + (size_t)m1 * ld;             // [m1]
  const int tid = threadIdx.x;
  for (int j = tid; j <= m2; j += NT) cr[j] = a.cr[g * (m2 + 1) + j];
  __syncthreads();
  const double t = cr[0];
  if (a.iter >= a.itr1) {
    const double s = a.sprev[g];
    if (fabs((t - s) / t) < a.dd) {  // uniform
      if (tid == 0) {
        a.done[g] = 1;
        a.iters[g] = a.iter;
        mcls_count_done(a.n_active);
      }
      return;
    }
    if (tid == 0) a.sprev[g] = t;
  }
  for (int idx = tid; idx < m1 * (m1 + 1); idx += NT) {
    const int i = idx / (m1 + 1), k = idx - i * (m1 + 1);
    double v;
    if (k == m1) {
      v = cr[i] - pow(-a.alpha, (double)i);
    } else {
      const int df = i > k ? i - k : k - i;
      double tv = cr[df];
      if (df == 0 || (df % 2 == 0)) tv += cr[0];
      double hv = cr[i + k];
      if (((i + k) & 1) == 0) hv -= cr[0];
      v = tv + hv;
    }
    A[i * ld + k] = v;
  }
  __syncthreads();
  for (int c = 0; c < m1; ++c) {
    const double piv = A[c * ld + c];
    for (int r = c + 1 + tid; r < m1; r += NT) fcol[r] = A[r * ld + c] / piv;
    __syncthreads();
    const int w = m1 - c;
    for (int idx = tid; idx < (m1 - 1 - c) * w; idx += NT) {
      const int r = c + 1 + idx / w, k = c + 1 + idx % w;
      A[r * ld + k] -= fcol[r] * A[c * ld + k];
    }
    __syncthreads();
  }
  if (tid < 64) {
    for (int r = m1 - 1; r >= 0; --r) {
      double s = 0.0;
      for (int k = r + 1 + tid; k < m1; k += 64) s += A[r * ld + k] * fcol[k];
      s = wave_sum(s);
      if (tid == 0) fcol[r] = (A[r * ld + m1] - s) / A[r * ld + r];
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int j = tid; j < m1; j += NT) a.mc[g * m1 + j] += fcol[j];
  if (tid == 0 && a.iter == a.itr2) {
    a.done[g] = 1;
    a.iters[g] = a.itr2;
    mcls_count_done(a.n_active);
  }
}

// ---- the same Newton update, one WAVE per frame (orders up to 63) -------------------------------
// The system matrix (Toeplitz + Hankel of cr) is symmetric positive definite and small; the
// workgroup-per-frame elimination above spends its time in 2 x 60 block barriers and index
// arithmetic (77 us per solve).  Here lane r owns row r in REGISTERS and the wave runs a
// Gauss-Jordan elimination without a single barrier:
//   * rows above the pivot are eliminated too (Gauss-Jordan), so no U factor is stored and no
//     back substitution runs: lane r keeps its pivot d_r and ends with x_r = b_r / d_r;
//   * the pivot row is never broadcast lane by lane: the trailing block is symmetric, so
//     p[k] = A[c][k] = A[k][c] is the entry of column c that lane k holds.
// Rounds 2-4 published that column through LDS and read it back as uniform-address (broadcast) loads, one per
// product: the kernel was bound by the LDS return path (0.80 busy; a quarter of the entries through v_readlane
// and SGPR operands balanced it against the VALU at 0.69 / 0.63: 1 635 us per launch at the bench size).
// the value lane l (wave-uniform) holds, in scalar registers
__device__ __forceinline__ double lane_value(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// What runs now keeps the pivot row in REGISTERS.  gfx90a+ can take a 64-bit VALU operand
// from lane n of the reader's own row of 16 lanes (DPP row_newbcast), so here the pivot row p[k] = A[k][c]
// (the leading elements of the lanes, by the symmetry of the trailing block) is laid out ONCE per step as four
// registers P_m, lane 16 rho + i holding p[16 m + i] in every row rho -- one 8-byte store and at most four
// 8-byte loads per lane through the wave's 512 bytes of LDS instead of W - c broadcasts -- and a product is one
// v_fmac_f64_dpp: row[k] += P_{k / 16}[lane k % 16 of the row] * (-f).  The rows are not shifted (the step loop
// is unrolled: every register index and broadcast lane is an immediate).  Per element the same multiply-adds
// with the same operands in the same order as in the LDS-broadcast kernel it replaced (the same bits with an IEEE
// division for 1 / pivot: 1 156 us per launch; with the reciprocal below the multipliers may differ in the last place).
template <int N>
__device__ __forceinline__ void fmac_row_bcast(double& acc, const double p, const double nf) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(p), "v"(nf), "n"(N));
}

// columns LO .. W - 1 of every row beyond column C: one product each
template <int W, int C, int LO, int... Ks>
__device__ __forceinline__ void ls_dpp_update(double (&row)[W], const double (&Pm)[4], const double nf,
                                              std::integer_sequence<int, Ks...>) {
  (((Ks > C && Ks >= LO) ? fmac_row_bcast<(Ks & 15)>(row[Ks], Pm[Ks >> 4], nf) : (void)0), ...);
}

// the pivot row of step C: the leading elements row[C] of all lanes, laid out for the broadcasts, and the pivot
template <int W, int C>
__device__ __forceinline__ void ls_dpp_fetch(const double (&row)[W], double* lead, int lane, double (&Pm)[4], double& pc) {
  lead[lane] = row[C];
  __builtin_amdgcn_wave_barrier();
  constexpr int M0 = (C + 1) >> 4;              // first register of the pivot row with a column beyond C
#pragma unroll
  for (int mm = 0; mm < 4; ++mm)
    if (mm >= M0 && 16 * mm < W) Pm[mm] = lead[16 * mm + (lane & 15)];
  pc = lead[C];                                 // pivot A[C][C] (uniform address)
  __builtin_amdgcn_wave_barrier();
}

// Step C with the pivot row Pm / pivot pc already in registers; the NEXT step's are fetched behind this step's
// first product (column C + 1 is final then), so the trip through LDS runs under the remaining products.
template <int W, int C>
__device__ __forceinline__ void ls_dpp_steps(double (&row)[W], double& b, double& d, double* lead, int lane, int m1,
                                             const double (&Pm)[4], const double pc) {
  if constexpr (C < W) {
    if (C < m1) {                                   // wave-uniform
      const double bc = lane_value(b, C);
      const bool is_piv = lane == C;
      if (is_piv) d = pc;
      // 1 / pivot (positive, normal; wave-uniform): the hardware's estimate and two Newton steps -- within an ulp,
      // five instructions instead of the eleven of the IEEE division, which were a seventh of a step
      double inv = __builtin_amdgcn_rcp(pc);
      inv = fma(inv, fma(-pc, inv, 1.0), inv);
      inv = fma(inv, fma(-pc, inv, 1.0), inv);
      const double f = is_piv ? 0.0 : row[C] * inv;
      const double nf = -f;
      // (a VALU write of a DPP source needs two wait states before the DPP read; the compiler cannot see into
      // the asm statements, so the distance is put here once per step -- the P registers come from LDS loads,
      // but a register copy in front of the sequence would be a VALU write)
      asm volatile("s_nop 1" ::: "memory");
      double Pn[4], pcn = 1.0;        // (only the registers with columns beyond C + 1 are loaded, and only those are used)
      if constexpr (C + 1 < W) {
        fmac_row_bcast<((C + 1) & 15)>(row[C + 1], Pm[(C + 1) >> 4], nf);
        ls_dpp_fetch<W, C + 1>(row, lead, lane, Pn, pcn);      // (unconditional: one idle fetch behind the last step)
      }
      ls_dpp_update<W, C, C + 2>(row, Pm, nf, std::make_integer_sequence<int, W>{});
      b -= f * bc;
      ls_dpp_steps<W, C + 1>(row, b, d, lead, lane, m1, Pn, pcn);
    }
  }
}

template <int W>   // W >= m + 1, W <= 64
__global__ __launch_bounds__(256, 3) void mcls_solve_dpp_kernel(LsArgs a) {
  // cr[0 .. 2m] of the wave's frame with the reference's parity terms applied once (theq's matrix is
  // A[r][k] = (cr[|r - k|] (+ t)) + (cr[r + k] (- t)), the t's where the index is even, t = cr[0]): 119 entries
  // prepared per solve instead of 3 600 parity tests
  __shared__ double crp[4][2 * 64];   // cr[j] + t [j even]
  __shared__ double crm[4][2 * 64];   // cr[j] - t [j even]
  __shared__ double leads[4][64];     // the leading elements of the step (the pivot row)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t slot = (int64_t)blockIdx.x * 4 + wv;
  if (slot >= a.n_rows) return;                      // wave-uniform
  const int64_t g = a.rows ? a.rows[slot] : slot;
  if (a.done[g]) return;
  const int m = a.m, m1 = m + 1, m2 = 2 * m;
  const double* crg = a.cr + g * (m2 + 1);
  const double t = crg[0];
  if (a.iter >= a.itr1) {
    const double sp = a.sprev[g];
    if (fabs((t - sp) / t) < a.dd) {                 // uniform
      if (lane == 0) {
        a.done[g] = 1;
        a.iters[g] = a.iter;
        mcls_count_done(a.n_active);
      }
      return;
    }
    if (lane == 0) a.sprev[g] = t;
  }
  for (int j = lane; j <= m2; j += 64) {
    const double c = crg[j];
    const bool even = (j & 1) == 0;
    crp[wv][j] = even ? c + t : c;
    crm[wv][j] = even ? c - t : c;
  }
  __builtin_amdgcn_wave_barrier();
  const int r = lane;
  const bool rowok = r < m1;
  double row[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    double v = 0.0;
    if (rowok && k < m1) v = crp[wv][r > k ? r - k : k - r] + crm[wv][r + k];
    row[k] = v;
  }
  double b = rowok ? crg[r] - a.apow[r] : 0.0;     // (the library's pow per lane and solve was 150 instructions)
  double d = 1.0;
  double P0[4], pc0 = 1.0;
  ls_dpp_fetch<W, 0>(row, leads[wv], lane, P0, pc0);
  ls_dpp_steps<W, 0>(row, b, d, leads[wv], lane, m1, P0, pc0);
  if (rowok) a.mc[g * m1 + r] += b / d;
  if (lane == 0 && a.iter == a.itr2) {
    a.done[g] = 1;
    a.iters[g] = a.itr2;
    mcls_count_done(a.n_active);
  }
}

__global__ void mcls_alpha_pow_kernel(double alpha, int m1, double* __restrict__ apow) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < m1) apow[r] = pow(-alpha, (double)r);
}

__global__ void mcls_finalize_kernel(const double* __restrict__ mc, int64_t T, int m1,
                                     float* __restrict__ o32, int64_t ld32, double* __restrict__ o64,
                                     const int* __restrict__ iters_in, int* __restrict__ iters_out) {
  const int64_t n = T * m1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / m1;
    const int j = (int)(i - t * m1);
    const double v = mc[i];
    if (o32) o32[t * ld32 + j] = (float)v;
    if (o64) o64[i] = v;
    if (iters_out && j == 0) iters_out[t] = iters_in[t];
  }
}

int launch_gemm_f64_mgc2sp(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t T, int N, int K,
                           float* o32, double* o64, double* opow, hipStream_t s) {
  if (T <= 0) return ITTS_OK;
  ITTS_REQUIRE(K <= 64, "launch_gemm_f64_mgc2sp: K <= 64");
  const bool vec = lda % 2 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && K % 4 == 0;
  const unsigned grid = gemm_f64_grid(T, N);
  const GemmOut out{o32, o64, opow};
  if (vec)
    hipLaunchKernelGGL((gemm_f64_kernel<true, false, true>), dim3(grid), dim3(256), 0, s, A, lda, B, ldb,
                       (double*)nullptr, (int64_t)0, T, N, K, (const int*)nullptr, (const double*)nullptr, out);
  else
    hipLaunchKernelGGL((gemm_f64_kernel<false, false, true>), dim3(grid), dim3(256), 0, s, A, lda, B, ldb,
                       (double*)nullptr, (int64_t)0, T, N, K, (const int*)nullptr, (const double*)nullptr, out);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

// rows of C (the active ones): C = aux / exp(2 C), k < N -- the same expression as the RATIO epilogue of
// gemm_f64_kernel, for the products that do not run on that kernel (order > 63)
__global__ void mcls_ratio_kernel(double* __restrict__ C, const double* __restrict__ aux, int64_t ldc, int64_t T,
                                  int N, const int* __restrict__ rows) {
  const int64_t n = T * N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / N;
    const int k = (int)(i - r * N);
    const int64_t g = rows ? rows[r] : r;
    C[g * ldc + k] = aux[g * ldc + k] / exp(2.0 * C[g * ldc + k]);
  }
}

// C = aux / exp(2 A B): launch_gemm_f64 with the ratio formed in the epilogue where the kernel allows it
int launch_gemm_f64_ratio(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                          int64_t T, int N, int K, const int* rows, const double* aux, hipStream_t s) {
  if (T <= 0) return ITTS_OK;
  const bool vec = lda % 2 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && K % 4 == 0;
  if (K <= 64) {
    const unsigned grid = gemm_f64_grid(T, N);
    if (vec)
      hipLaunchKernelGGL((gemm_f64_kernel<true, true>), dim3(grid), dim3(256), 0, s, A, lda, B, ldb, C, ldc, T, N, K, rows, aux);
    else
      hipLaunchKernelGGL((gemm_f64_kernel<false, true>), dim3(grid), dim3(256), 0, s, A, lda, B, ldb, C, ldc, T, N, K, rows, aux);
    ITTS_LAUNCH_CHECK();
    return ITTS_OK;
  }
  const int rc = launch_gemm_f64(A, lda, B, ldb, C, ldc, T, N, K, rows, s, false);
  if (rc) return rc;
  hipLaunchKernelGGL(mcls_ratio_kernel, dim3((unsigned)std::min<int64_t>((T * N + 255) / 256, 16384)), dim3(256), 0, s,
                     C, aux, ldc, T, N, rows);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

int launch_gemm_f64(const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                    int64_t ldc, int64_t T, int N, int K, const int* rows, hipStream_t s,
                    bool a_has_slack) {
  if (T <= 0) return ITTS_OK;
  dim3 grid((unsigned)((T + 127) / 128), (unsigned)((N + 63) / 64));
  // the 16-byte loads of A cover k .. k+3: past the end of a row (and of the last row's buffer)
  // unless K is a multiple of 4 or the caller's buffer has that slack
  const bool vec = lda % 2 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (a_has_slack || K % 4 == 0);
  if (K > 64 && T >= 1024) {      // long K, enough rows to fill the chip: the staged kernel
    static bool attr_set = false;
    if (!attr_set) {
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f64_staged_kernel<true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f64_staged_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      attr_set = true;
    }
    if (vec) {
      static bool attr2 = false;
      if (!attr2) {
        ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f64_lds_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        attr2 = true;
      }
      hipLaunchKernelGGL(gemm_f64_lds_kernel, dim3(grid.x * grid.y), dim3(256), GEMM_LDS_BYTES, s, A, lda, B, ldb, C, ldc, T, N, K, rows);
    } else if (vec)
      hipLaunchKernelGGL(gemm_f64_staged_kernel<true>, grid, dim3(256), 65536, s, A, lda, B, ldb, C, ldc, T, N, K, rows);
    else
      hipLaunchKernelGGL(gemm_f64_staged_kernel<false>, grid, dim3(256), 65536, s, A, lda, B, ldb, C, ldc, T, N, K, rows);
  } else if (vec)
    hipLaunchKernelGGL(gemm_f64_kernel<true>, dim3(gemm_f64_grid(T, N)), dim3(256), 0, s, A, lda, B, ldb, C, ldc, T, N, K, rows);
  else
    hipLaunchKernelGGL(gemm_f64_kernel<false>, dim3(gemm_f64_grid(T, N)), dim3(256), 0, s, A, lda, B, ldb, C, ldc, T, N, K, rows);
  ITTS_LAUNCH_CHECK();
  return ITTS_OK;
}

// d_in: amplitude or power spectra [T, K]; results as in itts_mcep.
int mcep_lockstep(DeviceContext* ctx, const double* d_in, int in_is_power, int64_t T, int K, int order,
                  double alpha, double eps, int miniter, int maxiter, double threshold, float* d_mc_f32,
                  int64_t ld_mc, double* d_mc_f64, int* d_iters, hipStream_t s) {
  const int flng = (K - 1) * 2, f2 = flng / 2, m1 = order + 1, m2 = 2 * order;
  const FreqtTables* ft = get_freqt(ctx, order, f2, alpha, true);
  if (!ft) return ITTS_E_HIP;
  int logflng = 0;
  while ((1 << logflng) < flng) ++logflng;
  double *xp = nullptr, *cbuf = nullptr, *mc = nullptr, *cr = nullptr, *sprev = nullptr;
  int *done = nullptr, *iters = nullptr, *n_active = nullptr, *rows = nullptr;
  const int64_t Kp = K + (K & 1);   // even pitch: 16-byte aligned rows for the vector loads
  // the vector loads may run up to 3 doubles past a row's end; the fused kernel's operand loads of the last
  // mel-cepstrum row up to 64 values (they meet zero rows of specP / crP: the slack is cleared, an Inf or NaN
  // would survive the zero)
  const size_t slack = 512;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&xp, (size_t)T * Kp * 8 + slack, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&cbuf, (size_t)T * Kp * 8 + slack, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&mc, (size_t)T * m1 * 8 + slack, s));
  ITTS_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char*>(xp) + (size_t)T * Kp * 8, 0, slack, s));
  ITTS_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char*>(mc) + (size_t)T * m1 * 8, 0, slack, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&cr, (size_t)T * (m2 + 1) * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&sprev, (size_t)T * 8, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&done, (size_t)T * 4, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&iters, (size_t)T * 4, s));
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&rows, (size_t)T * 4, s));
  // [MCLS_COUNTERS x 32] frames done, then [0] frames still iterating (mcls_remaining_kernel), [1] list cursor
  constexpr size_t kCountInts = (size_t)MCLS_COUNTERS * 32;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&n_active, (kCountInts + 2) * 4, s));
  double* apow = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&apow, (size_t)m1 * 8, s));
  hipLaunchKernelGGL(mcls_alpha_pow_kernel, dim3((m1 + 63) / 64), dim3(64), 0, s, alpha, m1, apow);
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(hipMemsetAsync(n_active, 0, (kCountInts + 2) * 4, s));
  int* n_left = n_active + kCountInts;
  LsArgs a{};
  a.in = d_in; a.in_is_power = in_is_power; a.T = T; a.flng = flng; a.logflng = logflng; a.m = order;
  a.alpha = alpha; a.eps = eps; a.itr1 = miniter; a.itr2 = maxiter; a.dd = threshold; a.xp = xp;
  a.cbuf = cbuf; a.mc = mc; a.cr = cr; a.sprev = sprev; a.done = done; a.iters = iters;
  a.n_active = n_active; a.ldk = Kp; a.rows = nullptr; a.n_rows = T; a.apow = apow;
  const size_t lds_solve = (size_t)(m2 + 2 + (size_t)m1 * (order + 2) + m1 + 2) * 8;
  ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mcls_solve_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_solve));
  // both products of a round in one kernel where its accumulators fit (order <= 63), and the start of the loop in
  // one kernel likewise; ITTS_MCEP_FUSED=0: the separate launches
  const char* fenv = getenv("ITTS_MCEP_FUSED");
  const bool fused = m1 <= 64 && m2 + 1 <= 128 && ft->specP && ft->crP && ft->initP && !(fenv && fenv[0] == '0');
  int rc = ITTS_OK;
  if (fused) {
    static std::atomic<uint64_t> iattr{0};
    int dev = 0;
    ITTS_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !((iattr.load() >> dev) & 1)) {
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mcls_init_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FI_LDS_BYTES));
      if (dev < 64) iattr.fetch_or(uint64_t(1) << dev);
    }
    hipLaunchKernelGGL(mcls_init_fused_kernel, dim3((unsigned)((T + 16 * FI_WAVES - 1) / (16 * FI_WAVES))), dim3(64 * FI_WAVES),
                       FI_LDS_BYTES, s, a, ft->initP, ft->kpad);
    ITTS_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(mcls_init_flat_kernel, dim3((unsigned)std::min<int64_t>((T + 3) / 4, 8192)), dim3(256), 0, s, a);
    ITTS_LAUNCH_CHECK();
    if ((rc = launch_gemm_f64(cbuf, Kp, ft->initT, m1, mc, m1, T, m1, K, nullptr, s))) return rc;
  }
  int* rows_all = nullptr;          // the identity list the fused kernel walks while no frame has converged yet
  if (fused) {
    ITTS_HIP_CHECK(itts::scratch_malloc((void**)&rows_all, (size_t)T * 4, s));
    hipLaunchKernelGGL(mcls_iota_kernel, dim3((unsigned)std::min<int64_t>((T + 255) / 256, 2048)), dim3(256), 0, s, rows_all, T);
    ITTS_LAUNCH_CHECK();
  }
  if (fused) {
    static std::atomic<uint64_t> attr_done{0};
    int dev = 0;
    ITTS_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !((attr_done.load() >> dev) & 1)) {
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mcls_fused3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES));
      ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)mcls_fused3_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES));
      if (dev < 64) attr_done.fetch_or(uint64_t(1) << dev);
    }
  }
  for (int it = 1; it <= maxiter; ++it) {
    a.iter = it;
    const int64_t nr = a.n_rows;      // frames still iterating (their list is a.rows)
    // log spectrum of the current model and the ratio to the periodogram in one product (the two
    // transforms of the reference's loop are folded into the warping matrices: FreqtTables), then the
    // warped autocorrelation of the ratio
    if (fused) {
      const dim3 fgrid((unsigned)((nr + 16 * F3_WAVES - 1) / (16 * F3_WAVES))), fblock(64 * F3_WAVES);
      const int* rl = a.rows ? a.rows : rows_all;
      if (m2 + 1 <= 64)
        hipLaunchKernelGGL(mcls_fused3_kernel<4>, fgrid, fblock, F3_LDS_BYTES, s, mc, m1, ft->specP, xp, Kp, ft->crP, cr, nr, K,
                           ft->kpad, m2 + 1, rl);
      else
        hipLaunchKernelGGL(mcls_fused3_kernel<8>, fgrid, fblock, F3_LDS_BYTES, s, mc, m1, ft->specP, xp, Kp, ft->crP, cr, nr, K,
                           ft->kpad, m2 + 1, rl);
      ITTS_LAUNCH_CHECK();
    } else {
      if ((rc = launch_gemm_f64_ratio(mc, m1, ft->specT, K, cbuf, Kp, nr, K, m1, a.rows, xp, s))) return rc;
      if ((rc = launch_gemm_f64(cbuf, Kp, ft->crT, m2 + 1, cr, m2 + 1, nr, m2 + 1, K, a.rows, s))) return rc;
    }
    const dim3 wgrid((unsigned)((nr + 3) / 4));
    if (m1 <= 20) hipLaunchKernelGGL(mcls_solve_dpp_kernel<20>, wgrid, dim3(256), 0, s, a);
    else if (m1 <= 24) hipLaunchKernelGGL(mcls_solve_dpp_kernel<24>, wgrid, dim3(256), 0, s, a);
    else if (m1 <= 32) hipLaunchKernelGGL(mcls_solve_dpp_kernel<32>, wgrid, dim3(256), 0, s, a);
    else if (m1 <= 48) hipLaunchKernelGGL(mcls_solve_dpp_kernel<48>, wgrid, dim3(256), 0, s, a);
    else if (m1 <= 60) hipLaunchKernelGGL(mcls_solve_dpp_kernel<60>, wgrid, dim3(256), 0, s, a);
    else if (m1 <= 64) hipLaunchKernelGGL(mcls_solve_dpp_kernel<64>, wgrid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mcls_solve_kernel, dim3((unsigned)nr), dim3(NT), lds_solve, s, a);
    ITTS_LAUNCH_CHECK();
    if (it >= miniter && it < maxiter) {
      // the count comes back into a page-locked word the host spins on (a copy into pageable memory went through the
      // runtime's staging buffer: 31-39 us from the end of the solve to the next launch, eight times a call)
      int remaining = 0;
      {
        volatile int* w = reinterpret_cast<volatile int*>(pinned_slot(ctx));
        constexpr int kPending = 0x7fffffff;
        w[0] = kPending;
        hipLaunchKernelGGL(mcls_remaining_kernel, dim3(1), dim3(64), 0, s, n_active, (int)T, n_left);
        ITTS_LAUNCH_CHECK();
        ITTS_HIP_CHECK(hipMemcpyAsync(const_cast<int*>(w), n_left, 4, hipMemcpyDeviceToHost, s));
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; w[0] == kPending; ++spins) {
          __builtin_ia32_pause();
          if ((spins & 0xffff) == 0xffff &&
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.2) {
            ITTS_HIP_CHECK(hipStreamSynchronize(s));
            break;
          }
        }
        ITTS_REQUIRE(w[0] != kPending, "the frame count did not come back");
        remaining = w[0];
      }
      if (remaining <= 0) break;
      if (remaining < nr) {          // shrink the work list to the frames that still iterate
        ITTS_HIP_CHECK(hipMemsetAsync(n_left + 1, 0, 4, s));
        hipLaunchKernelGGL(mcls_compact_kernel, dim3((unsigned)std::min<int64_t>((T + 255) / 256, 1024)),
                           dim3(256), 0, s, done, T, rows, n_left + 1);
        ITTS_LAUNCH_CHECK();
        a.rows = rows;
        a.n_rows = remaining;
      }
    }
  }
  const int64_t n = T * m1;
  hipLaunchKernelGGL(mcls_finalize_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256),
                     0, s, mc, T, m1, d_mc_f32, ld_mc, d_mc_f64, iters, d_iters);
  ITTS_LAUNCH_CHECK();
  ITTS_HIP_CHECK(itts::scratch_free(xp, s));
  ITTS_HIP_CHECK(itts::scratch_free(cbuf, s));
  ITTS_HIP_CHECK(itts::scratch_free(mc, s));
  ITTS_HIP_CHECK(itts::scratch_free(cr, s));
  ITTS_HIP_CHECK(itts::scratch_free(sprev, s));
  ITTS_HIP_CHECK(itts::scratch_free(done, s));
  ITTS_HIP_CHECK(itts::scratch_free(iters, s));
  ITTS_HIP_CHECK(itts::scratch_free(n_active, s));
  ITTS_HIP_CHECK(itts::scratch_free(apow, s));
  ITTS_HIP_CHECK(itts::scratch_free(rows, s));
  if (rows_all) ITTS_HIP_CHECK(itts::scratch_free(rows_all, s));
  return ITTS_OK;
}

}  // namespace itts
