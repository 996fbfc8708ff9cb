// The sum of a set of non-negative doubles without its R largest, for a 256-thread workgroup that holds the
// set a few values per thread (D4C's coarse aperiodicity: d4c.cpp sorts each band's power spectrum and sums
// all but the boundary + 1 largest bins).  Used by d4c_kernel (world_f0ap.hip); scripts/select_lab checks it
// against a sort on the host, ties included (tests/test_gpu_select.py).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace itts {

// Rounds 1-3 popped the R = boundary + 1 largest bins one by one with a workgroup-wide arg-max: two
// barriers and six lane-permute steps per round, 22 rounds at 16 kHz and 5 x 65 at 48 kHz -- 2.2 of the
// kernel's 10.5 ms at 16 kHz and 41 of 67 ms at 48 kHz (profiles/r4za_d4c_topk_lab.txt).  Now:
//   1. every WAVE pops the largest of its own 256+ bins (a thread's bins are sorted, its head is the
//      candidate; the wave's maximum by DPP, no LDS, no barrier) into a sorted list in LDS -- R / 3 at a
//      time, until no wave's list is admitted whole;
//   2. the R largest of the frame are among those 4 R: a candidate's rank in the union is its position in
//      its own list plus, by binary search, the entries of the other lists ahead of it (ties: lower wave
//      first) -- the first n_w entries of wave w's list make the cut;
//   3. every thread sums its bins below its wave's last admitted value; equal values are counted, not
//      identified.
// One or two barriers per batch and one for the sum.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)u, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(u >> 32), (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lane_value64(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// maximum over the wave, the same value in every lane (values >= -1, or NaN everywhere)
__device__ __forceinline__ double wave_max_dpp(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));       // quad_perm [1, 0, 3, 2]
  v = fmax(v, dpp_f64<0x4E>(v));       // quad_perm [2, 3, 0, 1]
  v = fmax(v, dpp_f64<0x141>(v));      // row_half_mirror
  v = fmax(v, dpp_f64<0x140>(v));      // row_mirror: every lane holds the maximum of its row of 16
  return fmax(fmax(lane_value64(v, 0), lane_value64(v, 16)), fmax(lane_value64(v, 32), lane_value64(v, 48)));
}

// uniform lane index -> that lane's value
__device__ __forceinline__ double lane_value64u(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// mine[0 .. cnt): this thread's bins, sorted descending (fillers -1 behind them).  lists: 4 R + 4 doubles of
// LDS nobody else uses until the call returns; red: >= 4 doubles.  Returns the sum in every thread.
//
// Round 5: a wave no longer pops its largest bins ONE a round.  Every head that is larger than every lane's
// SECOND bin is larger than everything that is not a head, so all such heads are the wave's next largest at
// once: one round takes them all, ranks them among themselves (a loop over the set lanes of the ballot) and
// appends them to the list in order.  A band's large bins are neighbours in frequency -- one to a lane -- so
// its 22 (16 kHz) or 65 (48 kHz) largest go in one or two rounds per wave instead of 22 or 65; when no head
// is strictly larger than the largest second bin (ties, or a lane that owns two of the largest) the round
// pops one bin as before.  Waves now hold lists of different lengths (counts behind the lists).
template <int MPER>
__device__ __forceinline__ double d4c_rest_without_largest(const double (&mine)[MPER], int cnt, int R, double* lists,
                                                           double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* own = lists + wv * R;
  int* counts = reinterpret_cast<int*>(lists + 4 * R);      // entries in each wave's list
  // 1. + 2.  For large R in steps of about R / 3 entries: a wave whose list is not admitted whole has
  // nothing further to offer (its next bins are no larger, the lists only grow), so the rounds stop as
  // soon as that holds for all four -- after R / 3 or 2 R / 3 entries for all but freak spectra.
  const int batch = R <= 32 ? R : (R + 2) / 3;
  int popped = 0, K = 0, n_w = 0, target = 0;
  double head = mine[0];                       // -1 when the thread has no bin
  for (;;) {
    target = target + batch < R ? target + batch : R;
    while (K < target) {
      if (__ballot(popped < cnt) == 0ull) break;             // the wave has nothing left
      double second = -1.0;
#pragma unroll
      for (int i = 1; i < MPER; ++i)
        if (i == popped + 1 && i < cnt) second = mine[i];
      const double m2 = wave_max_dpp(second);
      const bool cand = popped < cnt && head > m2;
      const unsigned long long mask = __ballot(cand);
      if (mask == 0ull) {
        // the largest head ties with a second bin: one bin this round
        const double m = wave_max_dpp(head);
        const unsigned long long who = __ballot(head == m && popped < cnt);
        if (lane == __ffsll((long long)who) - 1) {
          ++popped;
          head = -1.0;
#pragma unroll
          for (int i = 1; i < MPER; ++i)
            if (i == popped && i < cnt) head = mine[i];
        }
        if (lane == 0) own[K] = m;
        K += 1;
      } else {
        int rank = 0;
        for (unsigned long long mm = mask; mm != 0ull; mm &= mm - 1ull) {
          const int bl = __ffsll((long long)mm) - 1;
          const double hb = lane_value64u(head, bl);
          rank += (hb > head || (hb == head && bl < lane)) ? 1 : 0;
        }
        if (cand) {
          if (K + rank < R) own[K + rank] = head;
          ++popped;
          head = -1.0;
#pragma unroll
          for (int i = 1; i < MPER; ++i)
            if (i == popped && i < cnt) head = mine[i];
        }
        const int n = __popcll(mask);
        K = K + n < R ? K + n : R;
      }
    }
    if (lane == 0) counts[wv] = K;
    __syncthreads();
    // how many of this wave's K entries are among the R largest of all the lists
    n_w = 0;
    for (int r0 = 0; r0 < K; r0 += 64) {
      const int r = r0 + lane;
      bool in_cut = false;
      if (r < K) {
        const double c = own[r];
        int rank = r;
#pragma unroll
        for (int o = 1; o < 4; ++o) {
          const int w2 = (wv + o) & 3;
          const double* other = lists + w2 * R;
          const int Ko = counts[w2];
          int lo = 0, hi = Ko;
#pragma unroll
          for (int it = 0; it < 7; ++it) {             // Ko <= R <= 66 < 128
            const int mid = (lo + hi) >> 1;
            const double e = Ko > 0 ? other[mid < Ko ? mid : Ko - 1] : -1.0;
            const bool ahead = e > c || (e == c && w2 < wv);
            if (lo < hi) {
              if (ahead) lo = mid + 1; else hi = mid;
            }
          }
          rank += lo;
        }
        in_cut = c >= 0.0 && rank < R;
      }
      n_w += __popcll(__ballot(in_cut));
    }
    if (target == R) break;
    // a wave whose list went in whole and that has more to offer must go on
    const bool more = n_w == K && K < R && __ballot(popped < cnt) != 0ull;
    if (!__syncthreads_or(more)) break;      // (also: every wave has read the lists before the next batch extends them)
  }
  // 3. this wave's bins outside the cut
  const double v_last = n_w > 0 ? own[n_w - 1] : 0.0;
  double s_lt = 0.0;
  int c_gt = 0, c_eq = 0;
#pragma unroll
  for (int i = MPER - 1; i >= 0; --i) {
    if (i < cnt) {
      if (n_w == 0 || mine[i] < v_last) s_lt += mine[i];
      else if (mine[i] > v_last) ++c_gt;
      else ++c_eq;
    }
  }
  int packed = c_gt | (c_eq << 16);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) packed += __shfl_xor(packed, off, 64);
  s_lt = wave_sum(s_lt);
  const int n_gt = packed & 0xffff, q_eq = packed >> 16;
  const double rest_w = s_lt + (double)(q_eq - (n_w - n_gt)) * v_last;
  if (lane == 0) red[wv] = rest_w;
  __syncthreads();
  const double rest = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();                             // red and the lists are free again
  return rest;
}


}  // namespace itts
