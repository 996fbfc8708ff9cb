// Library-level C-ABI entry points: error reporting and the integer / scalar helpers the
// reference obtains from pyworld / pysptk (AudioProcessing.py:32-71).
#include <cmath>
#include <mutex>
#include <vector>

#include "common.h"

namespace itts {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace itts

extern "C" int itts_abi_version(void) { return 1; }

extern "C" const char* itts_last_error(void) { return itts::g_last_error.c_str(); }

extern "C" int itts_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// WORLD GetFFTSizeForCheapTrick: 2^(1 + floor(log2(3 fs / f0_floor + 1)))
extern "C" int itts_cheaptrick_fft_size(int fs, double f0_floor) {
  if (fs <= 0 || f0_floor <= 0) return ITTS_E_INVALID;
  return (int)std::pow(2.0, 1.0 + (int)(std::log(3.0 * fs / f0_floor + 1) / std::log(2.0)));
}

// WORLD GetNumberOfAperiodicities: floor(min(15000, fs/2 - 3000) / 3000)
extern "C" int itts_num_aperiodicities(int fs) {
  if (fs <= 0) return ITTS_E_INVALID;
  return (int)(std::min(15000.0, fs / 2.0 - 3000.0) / 3000.0);
}

// pysptk.util.mcepalpha(fs, start=0, stop=1, step=0.001, num_points=1000): the all-pass constant
// whose phase response best matches the mel scale in the RMS sense.
extern "C" double itts_mcep_alpha(int fs) {
  if (fs <= 0) return -1.0;
  const int n = 1000;
  std::vector<double> mel(n), warp(n);
  const double step_hz = (fs / 2.0) / n;
  for (int i = 0; i < n; ++i)
    mel[i] = 1000.0 / std::log(2.0) * std::log(1.0 + step_hz * i / 1000.0);
  const double mel_last = mel[n - 1];
  for (int i = 0; i < n; ++i) mel[i] /= mel_last;
  double best = 1e300, best_alpha = 0.0;
  const int n_cand = 1000;  // np.arange(0.0, 1.0, 0.001)
  for (int c = 0; c < n_cand; ++c) {
    const double alpha = 0.0 + c * 0.001;
    const double step = M_PI / n;
    for (int i = 0; i < n; ++i) {
      const double omega = step * i;
      const double num = (1 - alpha * alpha) * std::sin(omega);
      const double den = (1 + alpha * alpha) * std::cos(omega) - 2 * alpha;
      double w = std::atan(num / den);
      if (w < 0) w += M_PI;
      warp[i] = w;
    }
    const double last = warp[n - 1];
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
      const double d = mel[i] - warp[i] / last;
      acc += d * d;
    }
    const double dist = std::sqrt(acc / n);
    if (dist < best) {
      best = dist;
      best_alpha = alpha;
    }
  }
  return best_alpha;
}

extern "C" int64_t itts_world_num_frames(int64_t n_samples, int fs, double frame_period_ms) {
  if (n_samples < 0 || fs <= 0 || frame_period_ms <= 0) return ITTS_E_INVALID;
  return (int64_t)(1000.0 * n_samples / fs / frame_period_ms) + 1;
}

extern "C" int64_t itts_world_synth_length(int64_t n_frames, int fs, double frame_period_ms) {
  if (n_frames < 0 || fs <= 0 || frame_period_ms <= 0) return ITTS_E_INVALID;
  return (int64_t)(n_frames * frame_period_ms * fs / 1000.0);
}

// pyworld.wav2world(x, fs, fft_size, frame_period) (WorldFeatLabelGen.py:792-793) in one call:
// DIO -> StoneMask -> CheapTrick -> D4C on utterances stored back to back.
extern "C" int itts_wav2world(const double* d_x, const int64_t* h_x_off, const int64_t* h_f_off,
                              int n_utts, int fs, double frame_period_ms, int fft_size, double* d_f0,
                              double* d_sp, double* d_ap, void* stream) {
  ITTS_REQUIRE(h_x_off && h_f_off && n_utts >= 0, "null offsets");
  if (n_utts == 0 || h_f_off[n_utts] == 0) return ITTS_OK;
  ITTS_REQUIRE(d_x && d_f0 && (d_sp || d_ap), "null pointer");
  if (fft_size <= 0) fft_size = itts_cheaptrick_fft_size(fs, 71.0);
  hipStream_t s = itts::as_stream(stream);
  itts::ScratchScope scratch_scope(s);
  double* d_f0_raw = nullptr;
  ITTS_HIP_CHECK(itts::scratch_malloc((void**)&d_f0_raw, (size_t)h_f_off[n_utts] * sizeof(double), s));
  int rc = itts_dio(d_x, h_x_off, h_f_off, n_utts, fs, frame_period_ms, 71.0, 800.0, 2.0, 0.1, d_f0_raw,
                    stream);
  if (rc == ITTS_OK)
    rc = itts_stonemask(d_x, h_x_off, d_f0_raw, h_f_off, n_utts, fs, frame_period_ms, d_f0, stream);
  if (rc == ITTS_OK && d_sp)
    rc = itts_cheaptrick_mcep(d_x, h_x_off, d_f0, h_f_off, n_utts, fs, frame_period_ms, fft_size, -0.15,
                              d_sp, 0, 0.0, 0.0, 0, 0, 0.0, nullptr, 0, nullptr, nullptr, stream);
  if (rc == ITTS_OK && d_ap)
    rc = itts_d4c(d_x, h_x_off, d_f0, h_f_off, n_utts, fs, frame_period_ms, fft_size, 0.85, d_ap, nullptr,
                  nullptr, 0, stream);
  ITTS_HIP_CHECK(itts::scratch_free(d_f0_raw, s));
  return rc;
}
