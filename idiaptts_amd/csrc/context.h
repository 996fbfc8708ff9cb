// Per-device read-only tables shared by the WORLD / SPTK kernels (twiddles, frequency-warping
// matrices).  Created lazily on the first GPU call of a device; never touched before that, so
// forked DataLoader workers do not inherit an initialised HIP runtime.
#pragma once
#include <cstdint>
#include <map>
#include <tuple>

#include "common.h"

namespace itts {

struct FreqtTables {
  // SPTK freqt / frqtr as matrices, stored input-major ("transposed") so that consecutive
  // lanes read consecutive outputs:
  double* fwdT = nullptr;  // [f2+1][m+1]   mc[j]  = sum_i fwdT[i][j] * c[i]      freqt(c, f2 -> m, +a)
  double* invT = nullptr;  // [m+1][f2+1]   c'[i]  = sum_j invT[j][i] * mc[j]    freqt(mc, m -> f2, -a)
  double* frqT = nullptr;  // [f2+1][2m+1]  cr[j]  = sum_i frqT[i][j] * r[i]     frqtr(r, f2 -> 2m, +a)
  // The Newton loop of mcep never needs c' or r themselves, only the log spectrum Re FFT(c') and the
  // warped autocorrelation of IFFT(d): both transforms are linear, so they are folded into the
  // warping matrices (round 4; the loop lost its two FFTs per frame and iteration):
  double* specT = nullptr; // [m+1][f2+1]   S[k]   = sum_j specT[j][k] * mc[j]    = Re rfft(freqt(mc, -a))[k]
  double* crT = nullptr;   // [f2+1][2m+1]  cr[j]  = sum_k crT[k][j] * d[k]       = frqtr(irfft(d))[j], d real
  // zero-padded copies for mcls_fused_products_kernel (m <= 63): every operand load a whole, aligned 32-byte vector
  double* specP = nullptr; // [64][kpad]    specT with rows m+1 .. 63 and columns f2+1 .. kpad-1 zero; kpad = f2+1 rounded up to 64
  double* crP = nullptr;   // [kpad][128]   crT with rows f2+1 .. kpad-1 and columns 2m+1 .. 127 zero
  double* initP = nullptr; // [kpad][64]    initT with rows f2+1 .. kpad-1 and columns m+1 .. 63 zero
  int kpad = 0;
  double* initT = nullptr; // [f2+1][m+1]   mc0[j] = sum_k initT[k][j] * lg[k]    = freqt(c, +a), c = irfft(lg) with c[0], c[f2] halved
  // SPTK mgcep's own transform b2c (freqt without the `+ a d[0]` in the zeroth term):
  double* b1T = nullptr;   // [m+1][f2+1]   c[i]   = sum_j b1T[j][i] * b[j]      b2c(b, m -> f2, -a)
  double* p2T = nullptr;   // [f2+1][2m+1]  p~[j]  = sum_i p2T[i][j] * p[i]      b2c(p, f2 -> 2m, +a)
  int m = 0, f2 = 0;
  double alpha = 0;
};

// WORLD's randn() is a xorshift128 stream (fixed seed per CheapTrick / Synthesis call) consumed in
// order; the kernels generate it out of order: one lane produces RNG_CHUNK consecutive normals
// from a state jumped ahead with GF(2) matrices B^(2^k), B = 12 * RNG_CHUNK generator steps.
constexpr int RNG_CHUNK = 64;
constexpr int RNG_NJUMP = 26;    // streams of up to RNG_CHUNK << RNG_NJUMP = 2^32 normals
constexpr int RNG_TABLE_LOG2 = 20;  // generator states of the first 2^20 chunks are tabulated (16 MB)
struct JumpTable {
  uint32_t col[RNG_NJUMP][128][4];  // column j of B^(2^k): image of basis vector e_j
  uint4* states;                    // [1 << RNG_TABLE_LOG2] state at the start of chunk c (seed fixed)
};

// generator state at the start of chunk `chunk` of a stream: table look-up for the low bits, GF(2)
// matrix-vector products for the (rarely set) high bits -- powers of B commute
__device__ __forceinline__ uint4 rng_chunk_state(const JumpTable* __restrict__ jt, int64_t chunk) {
  uint4 st = jt->states[chunk & ((1 << RNG_TABLE_LOG2) - 1)];
  for (int k = RNG_TABLE_LOG2; k < RNG_NJUMP; ++k) {
    if ((chunk >> k) & 1) {
      const uint32_t s[4] = {st.x, st.y, st.z, st.w};
      uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
      for (int j = 0; j < 128; ++j) {
        if ((s[j >> 5] >> (j & 31)) & 1u) {
          r0 ^= jt->col[k][j][0];
          r1 ^= jt->col[k][j][1];
          r2 ^= jt->col[k][j][2];
          r3 ^= jt->col[k][j][3];
        }
      }
      st = make_uint4(r0, r1, r2, r3);
    }
  }
  return st;
}

struct DeviceContext {
  int device = -1;
  double2* twiddles = nullptr;  // [TW_N/2] exp(+2 pi i k / TW_N)
  // the same values re-packed per transform size n = 2^L (L = 5 .. 14): tw_compact[L][k] =
  // twiddles[k * TW_N / n], k < n/2 -- contiguous, for kernels that read twiddles through the
  // cache instead of keeping a copy in LDS
  double2* tw_compact[16] = {nullptr};
  JumpTable* jump = nullptr;    // created by get_jump_table
  int64_t* pinned = nullptr;    // [64] page-locked host slots for small device -> host read-backs
  std::map<std::tuple<int, int, long long>, FreqtTables> freqt;
};

// Returns the context of the current device (creating it on first use); nullptr + error set on
// failure.
DeviceContext* get_context();
// Returns warping tables for (order m, f2 = fftlen/2, alpha); nullptr + error on failure.
// need_fwd_frq = false builds only invT; need_spec adds specT (mgc2sp's folded form; need_fwd_frq implies it).
const FreqtTables* get_freqt(DeviceContext* ctx, int m, int f2, double alpha, bool need_fwd_frq,
                             bool need_mgc = false, bool need_spec = false);

// C = A B as launch_gemm_f64, K <= 64, with mgc2sp's outputs formed in the epilogue instead of C: the raw
// value (o64), exp(float(value)) (o32) and its square as double (opow), any of them, rows of pitch N.
int launch_gemm_f64_mgc2sp(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t T, int N, int K,
                           float* o32, double* o64, double* opow, hipStream_t s);

// Jump matrices of the randn() generator on the context's device; nullptr + error on failure.
const JumpTable* get_jump_table(DeviceContext* ctx);
// R[off[u] + i] = the integer behind normal i of utterance u's stream (i < len[u]; off / len are
// device arrays): randn() = R / 2^28 - 6.  max_len bounds len[u] (sizes the grid).
int launch_randn_u32(DeviceContext* ctx, const int64_t* d_off, const int64_t* d_len, int n_utts,
                     int64_t max_len, uint32_t* d_R, hipStream_t s);

// C [T, N] = A [T, K] x B [K, N] in fp64 on the matrix cores (mcep_lockstep.hip); `rows` (or
// NULL) lists the rows of A / C to work on; a_has_slack: A may be read up to 3 doubles past a row.
int launch_gemm_f64(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                    int64_t T, int N, int K, const int* rows, hipStream_t s, bool a_has_slack = true);

// One of the context's page-locked host slots (round robin) for a read-back of a few bytes.
int64_t* pinned_slot(DeviceContext* ctx);

// Host -> device copy on stream s through a ring of page-locked staging buffers: the host returns
// as soon as the bytes are in the staging buffer (a copy from pageable memory makes the runtime
// stage it synchronously, which holds back the launches queued behind it).  `src` may be reused at
// once.
int staged_upload(void* d_dst, const void* src, size_t bytes, hipStream_t s);

// A small host table the kernels of ONE launch sequence read in place: copied into a page-locked staging slot (the
// device reads host memory at the same address), no copy on the stream.  For a few loads per workgroup -- a table
// every thread walks belongs in device memory (staged_upload).  pinned_table_end after the last consumer has been
// queued on s, on every path (it keeps the slot from being reused before that work has run).
struct PinnedTable {
  void* p = nullptr;
  void* slot = nullptr;
};
int pinned_table_begin(const void* src, size_t bytes, PinnedTable* t);
int pinned_table_end(PinnedTable* t, hipStream_t s);

// Small stream-ordered device copy of a host int64 array (offsets). Caller frees with
// scratch_free on the same stream.
int upload_i64(const int64_t* h, int n, int64_t** d_out, hipStream_t s);

}  // namespace itts
