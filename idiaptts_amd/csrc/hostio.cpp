// Host-side file I/O of the feature-extraction loop (no GPU, no HIP): the two things
// WorldFeatLabelGen.gen_data does per utterance around the analysis --
//   AudioProcessing.get_raw   (audio/AudioProcessing.py:107-120): read a wav file, scale to [-1, 1],
//                             pre-emphasis raw[i] - p * raw[i-1]
//   LabelGen._save_to_npz     (data_preparation/LabelGen.py:63-101 via save_output,
//                             world/WorldFeatLabelGen.py:1121-1172): one `.npz` archive per stream
// -- done for a whole batch by a pool of plain threads, so the Python side makes one call per
// batch and the interpreter lock is never the bottleneck (np.savez on 8 Python threads is slower
// than on one).  The archives are what np.savez writes: a ZIP file of stored (uncompressed)
// `<key>.npy` members in npy format 1.0; np.load reads them back.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/idiaptts_amd.h"

namespace itts {
void set_error(const std::string& msg);
}

namespace {

// ------------------------------------------------------------------------------------ wav
struct WavInfo {
  int fs = 0;
  int channels = 0;
  int bits = 0;
  int format = 0;        // 1 = PCM integer, 3 = IEEE float
  int64_t data_pos = 0;  // file offset of the samples
  int64_t frames = 0;    // samples per channel
};

uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

bool parse_wav(FILE* f, WavInfo* w, std::string* err) {
  unsigned char h[12];
  if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) != 0 || memcmp(h + 8, "WAVE", 4) != 0) {
    *err = "not a RIFF/WAVE file";
    return false;
  }
  bool have_fmt = false;
  for (;;) {
    unsigned char ch[8];
    if (fread(ch, 1, 8, f) != 8) break;
    const uint32_t size = rd32(ch + 4);
    if (memcmp(ch, "fmt ", 4) == 0) {
      unsigned char b[40];
      const size_t n = size < sizeof(b) ? size : sizeof(b);
      if (n < 16 || fread(b, 1, n, f) != n) { *err = "short fmt chunk"; return false; }
      w->format = rd16(b);
      w->channels = rd16(b + 2);
      w->fs = (int)rd32(b + 4);
      w->bits = rd16(b + 14);
      if (w->format == 0xFFFE && n >= 26) w->format = rd16(b + 24);   // WAVE_FORMAT_EXTENSIBLE
      have_fmt = true;
      if (fseek(f, (long)(size - n + (size & 1)), SEEK_CUR) != 0) break;
    } else if (memcmp(ch, "data", 4) == 0) {
      if (!have_fmt) { *err = "data chunk before fmt chunk"; return false; }
      w->data_pos = ftell(f);
      const int bytes = w->bits / 8 * w->channels;
      if (bytes <= 0) { *err = "bad sample size"; return false; }
      w->frames = size / bytes;
      return true;
    } else {
      if (fseek(f, (long)(size + (size & 1)), SEEK_CUR) != 0) break;
    }
  }
  *err = "no data chunk";
  return false;
}

bool supported(const WavInfo& w) {
  if (w.channels != 1) return false;
  if (w.format == 1) return w.bits == 8 || w.bits == 16 || w.bits == 32;
  if (w.format == 3) return w.bits == 32 || w.bits == 64;
  return false;
}

// samples in [-1, 1] as float64 (scipy.io.wavfile.read + the scaling of AudioProcessing.read_wav),
// then raw[i] - p * raw[i-1] with the product and the difference rounded separately (numpy).
bool read_wav_into(const char* path, double preemphasis, int64_t expect, double* out, std::string* err) {
  FILE* f = fopen(path, "rb");
  if (!f) { *err = std::string("cannot open ") + path; return false; }
  WavInfo w;
  bool ok = parse_wav(f, &w, err) && supported(w);
  if (ok && w.frames != expect) { *err = "length changed between the two passes"; ok = false; }
  if (ok) {
    fseek(f, (long)w.data_pos, SEEK_SET);
    const size_t bytes = (size_t)w.frames * (w.bits / 8);
    std::vector<unsigned char> buf(bytes);
    if (fread(buf.data(), 1, bytes, f) != bytes) { *err = "short read"; ok = false; }
    if (ok) {
      const int64_t n = w.frames;
      if (w.format == 1 && w.bits == 16) {
        const int16_t* s = reinterpret_cast<const int16_t*>(buf.data());
        for (int64_t i = 0; i < n; ++i) out[i] = (double)s[i] / 32768.0;
      } else if (w.format == 1 && w.bits == 32) {
        const int32_t* s = reinterpret_cast<const int32_t*>(buf.data());
        for (int64_t i = 0; i < n; ++i) out[i] = (double)s[i] / 2147483648.0;
      } else if (w.format == 1 && w.bits == 8) {
        for (int64_t i = 0; i < n; ++i) out[i] = ((double)buf[i] - 128.0) / 128.0;
      } else if (w.bits == 32) {
        const float* s = reinterpret_cast<const float*>(buf.data());
        for (int64_t i = 0; i < n; ++i) out[i] = (double)s[i];
      } else {
        memcpy(out, buf.data(), bytes);
      }
      if (preemphasis != 0.0) {
        // this file is compiled with -ffp-contract=off: product and difference round separately
        for (int64_t i = n - 1; i >= 1; --i) out[i] = out[i] - preemphasis * out[i - 1];
      }
    }
  }
  fclose(f);
  if (!ok && err->empty()) *err = std::string("unsupported wav layout: ") + path;
  return ok;
}

template <typename F>
bool run_parallel(int n, int n_threads, F&& job, std::string* first_err) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n) n_threads = n > 0 ? n : 1;
  std::atomic<int> next{0};
  std::atomic<bool> failed{false};
  std::vector<std::string> errs(n_threads);
  auto worker = [&](int t) {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n || failed.load()) break;
      if (!job(i, &errs[t])) failed.store(true);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_threads; ++t) th.emplace_back(worker, t);
  worker(0);
  for (auto& x : th) x.join();
  if (failed.load())
    for (auto& e : errs)
      if (!e.empty()) { *first_err = e; break; }
  return !failed.load();
}

// ------------------------------------------------------------------------------------ npz
uint32_t crc_table[8][256];
std::atomic<bool> crc_ready{false};

void crc_init() {
  if (crc_ready.load()) return;
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    crc_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t)
      crc_table[t][i] = (crc_table[t - 1][i] >> 8) ^ crc_table[0][crc_table[t - 1][i] & 0xFF];
  crc_ready.store(true);
}

uint32_t crc32_update(uint32_t crc, const unsigned char* p, size_t n) {   // slicing-by-8
  crc = ~crc;
  while (n >= 8) {
    uint32_t a, b;
    memcpy(&a, p, 4);
    memcpy(&b, p + 4, 4);
    a ^= crc;
    crc = crc_table[7][a & 0xFF] ^ crc_table[6][(a >> 8) & 0xFF] ^ crc_table[5][(a >> 16) & 0xFF] ^
          crc_table[4][a >> 24] ^ crc_table[3][b & 0xFF] ^ crc_table[2][(b >> 8) & 0xFF] ^
          crc_table[1][(b >> 16) & 0xFF] ^ crc_table[0][b >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) crc = crc_table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return ~crc;
}

void put16(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x & 0xFF); v.push_back((x >> 8) & 0xFF); }
void put32(std::vector<unsigned char>& v, uint32_t x) { put16(v, x & 0xFFFF); put16(v, x >> 16); }

struct Member {
  std::string name;   // "<key>.npy"
  uint32_t crc, size, offset;
};

// one `<key>.npy` member: npy 1.0 header + rows x cols float32 taken from a strided matrix
void append_member(std::vector<unsigned char>& z, std::vector<Member>& dir, const std::string& key,
                   const float* src, int64_t ld, int64_t rows, int cols) {
  std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(rows) +
                     ", " + std::to_string(cols) + "), }";
  size_t total = 10 + dict.size() + 1;
  const size_t pad = (64 - total % 64) % 64;
  dict.append(pad, ' ');
  dict.push_back('\n');
  // The member is built where it stays: local header (CRC patched in at the end), npy header, then the rows straight
  // from the strided matrix, each row's CRC taken right behind its copy while it is in the cache.  (Until round 6 the
  // body was assembled in a buffer of its own -- value-initialised, filled, summed, copied into the archive: five passes
  // over every float on the 16 cores a box grants, where the writers are what a gen_data pass ends on.)
  const size_t body_size = 10 + dict.size() + (size_t)rows * cols * 4;
  Member m;
  m.name = key + ".npy";
  m.size = (uint32_t)body_size;
  m.offset = (uint32_t)z.size();
  put32(z, 0x04034b50u); put16(z, 20); put16(z, 0); put16(z, 0);
  put16(z, 0); put16(z, 0x21);                       // time 00:00:00, date 1980-01-01
  const size_t crc_at = z.size();
  put32(z, 0); put32(z, m.size); put32(z, m.size);
  put16(z, (uint32_t)m.name.size()); put16(z, 0);
  z.insert(z.end(), m.name.begin(), m.name.end());
  const size_t body_at = z.size();
  const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
  z.insert(z.end(), magic, magic + 8);
  put16(z, (uint32_t)dict.size());
  z.insert(z.end(), dict.begin(), dict.end());
  uint32_t crc = crc32_update(0, z.data() + body_at, z.size() - body_at);
  const size_t row_bytes = (size_t)cols * 4;
  for (int64_t r = 0; r < rows; ++r) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(src + r * ld);
    z.insert(z.end(), p, p + row_bytes);
    crc = crc32_update(crc, z.data() + z.size() - row_bytes, row_bytes);
  }
  m.crc = crc;
  for (int k = 0; k < 4; ++k) z[crc_at + k] = (unsigned char)((crc >> (8 * k)) & 0xFF);
  dir.push_back(m);
}

// Member names of an existing archive (central directory of a ZIP file without comment); false
// when the file is not there or not such an archive.
bool archive_members(const char* path, std::vector<std::string>* names) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  bool ok = false;
  unsigned char e[22];
  if (fseek(f, -22, SEEK_END) == 0 && fread(e, 1, 22, f) == 22 && rd32(e) == 0x06054b50u) {
    const uint32_t n = rd16(e + 10), cd_size = rd32(e + 12), cd_off = rd32(e + 16);
    std::vector<unsigned char> cd(cd_size);
    if (fseek(f, (long)cd_off, SEEK_SET) == 0 && fread(cd.data(), 1, cd_size, f) == cd_size) {
      size_t p = 0;
      ok = true;
      for (uint32_t i = 0; i < n && ok; ++i) {
        if (p + 46 > cd.size() || rd32(cd.data() + p) != 0x02014b50u) { ok = false; break; }
        const size_t nl = rd16(cd.data() + p + 28), xl = rd16(cd.data() + p + 30), cl = rd16(cd.data() + p + 32);
        if (p + 46 + nl > cd.size()) { ok = false; break; }
        names->emplace_back(reinterpret_cast<const char*>(cd.data() + p + 46), nl);
        p += 46 + nl + xl + cl;
      }
    }
  }
  fclose(f);
  return ok;
}

bool finish_archive(std::vector<unsigned char>& z, const std::vector<Member>& dir, const char* path,
                    std::string* err) {
  const uint32_t cd_off = (uint32_t)z.size();
  for (const Member& m : dir) {
    put32(z, 0x02014b50u); put16(z, 20); put16(z, 20); put16(z, 0); put16(z, 0);
    put16(z, 0); put16(z, 0x21);
    put32(z, m.crc); put32(z, m.size); put32(z, m.size);
    put16(z, (uint32_t)m.name.size()); put16(z, 0); put16(z, 0); put16(z, 0); put16(z, 0);
    put32(z, 0x81800000u);                           // regular file, rw-------
    put32(z, m.offset);
    z.insert(z.end(), m.name.begin(), m.name.end());
  }
  const uint32_t cd_size = (uint32_t)z.size() - cd_off;
  put32(z, 0x06054b50u); put16(z, 0); put16(z, 0);
  put16(z, (uint32_t)dir.size()); put16(z, (uint32_t)dir.size());
  put32(z, cd_size); put32(z, cd_off); put16(z, 0);
  const std::string tmp = std::string(path) + "_tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) { *err = "cannot create " + tmp; return false; }
  const bool ok = fwrite(z.data(), 1, z.size(), f) == z.size();
  if (fclose(f) != 0 || !ok) { *err = "short write to " + tmp; remove(tmp.c_str()); return false; }
  if (rename(tmp.c_str(), path) != 0) { *err = std::string("cannot rename to ") + path; return false; }
  return true;
}

}  // namespace

extern "C" int itts_wav_info(const char* h_path, int* fs, int64_t* n_samples) {
  if (!h_path || !fs || !n_samples) { itts::set_error("itts_wav_info: null pointer"); return ITTS_E_INVALID; }
  FILE* f = fopen(h_path, "rb");
  if (!f) { itts::set_error(std::string("itts_wav_info: cannot open ") + h_path); return ITTS_E_INVALID; }
  WavInfo w;
  std::string err;
  const bool ok = parse_wav(f, &w, &err);
  fclose(f);
  if (!ok) { itts::set_error(std::string("itts_wav_info: ") + err + ": " + h_path); return ITTS_E_INVALID; }
  if (!supported(w)) {
    itts::set_error(std::string("itts_wav_info: only mono PCM 8/16/32-bit and float wav files: ") + h_path);
    return ITTS_E_UNSUPPORTED;
  }
  *fs = w.fs;
  *n_samples = w.frames;
  return ITTS_OK;
}

extern "C" int itts_wav_read_batch(const char* const* h_paths, int n_files, const int64_t* h_offsets,
                                   double preemphasis, double* h_out, int n_threads) {
  if (n_files < 0 || (n_files > 0 && (!h_paths || !h_offsets || !h_out))) {
    itts::set_error("itts_wav_read_batch: null pointer");
    return ITTS_E_INVALID;
  }
  std::string err;
  const bool ok = run_parallel(n_files, n_threads, [&](int i, std::string* e) {
    return read_wav_into(h_paths[i], preemphasis, h_offsets[i + 1] - h_offsets[i], h_out + h_offsets[i], e);
  }, &err);
  if (!ok) { itts::set_error("itts_wav_read_batch: " + err); return ITTS_E_INVALID; }
  return ITTS_OK;
}

extern "C" int itts_write_feature_archives(const float* h_feat, int64_t ld, const int64_t* h_f_off,
                                           int n_utts, const char* const* h_paths, int n_streams,
                                           const int* h_col0, const int* h_width, const int* h_parts,
                                           const char* const* h_keys, int n_threads,
                                           unsigned char* h_needs_merge) {
  if (n_utts < 0 || n_streams < 0 ||
      (n_utts * n_streams > 0 && (!h_feat || !h_f_off || !h_paths || !h_col0 || !h_width || !h_parts || !h_keys))) {
    itts::set_error("itts_write_feature_archives: null pointer");
    return ITTS_E_INVALID;
  }
  for (int s = 0; s < n_streams; ++s)
    if (h_width[s] <= 0 || (h_parts[s] != 1 && h_parts[s] != 3) || h_col0[s] < 0 ||
        h_col0[s] + (int64_t)h_width[s] * h_parts[s] > ld) {
      itts::set_error("itts_write_feature_archives: bad stream layout");
      return ITTS_E_INVALID;
    }
  for (int u = 0; u < n_utts; ++u)
    if ((h_f_off[u + 1] - h_f_off[u]) * ld * 4 > (int64_t)3 << 30) {
      itts::set_error("itts_write_feature_archives: utterance too large for a plain ZIP archive");
      return ITTS_E_UNSUPPORTED;
    }
  crc_init();
  static const char* suffix[3] = {"", "_deltas", "_double_deltas"};
  std::string err;
  const bool ok = run_parallel(n_utts * n_streams, n_threads, [&](int job, std::string* e) {
    const int u = job / n_streams, s = job % n_streams;
    const char* path = h_paths[(int64_t)u * n_streams + s];
    if (h_needs_merge) {
      // an existing archive that holds members this call does not write keeps them: left to the
      // caller's merging path (LabelGen._save_to_npz semantics)
      h_needs_merge[job] = 0;
      std::vector<std::string> have;
      FILE* probe = fopen(path, "rb");
      if (probe) {
        fclose(probe);
        bool subset = archive_members(path, &have);
        for (size_t i = 0; subset && i < have.size(); ++i) {
          bool found = false;
          for (int p = 0; p < h_parts[s]; ++p)
            found = found || have[i] == std::string(h_keys[s]) + suffix[p] + ".npy";
          subset = found;
        }
        if (!subset) { h_needs_merge[job] = 1; return true; }
      }
    }
    const int64_t rows = h_f_off[u + 1] - h_f_off[u];
    const float* base = h_feat + h_f_off[u] * ld + h_col0[s];
    std::vector<unsigned char> z;
    std::vector<Member> dir;
    z.reserve((size_t)rows * h_width[s] * h_parts[s] * 4 + 1024);
    for (int p = 0; p < h_parts[s]; ++p)
      append_member(z, dir, std::string(h_keys[s]) + suffix[p], base + (int64_t)p * h_width[s], ld, rows,
                    h_width[s]);
    return finish_archive(z, dir, path, e);
  }, &err);
  if (!ok) { itts::set_error("itts_write_feature_archives: " + err); return ITTS_E_INVALID; }
  return ITTS_OK;
}

// ---- per-item normalisation of the data readers ---------------------------------------------------------------------
// out[r][c] = (float)(((double)x[r][c] - sub[c]) / div[c]): what numpy computes for `((sample - sub) / div)
// .astype(float32)` with a float32 sample and float64 parameters (NpzDataReader.preprocess_sample :347-371,
// QuestionLabelGen / WorldFeatLabelGen) -- the same IEEE operations in the same order, so the same bits -- without the
// two float64 temporaries of the whole sample: 14 ms -> 0.5 ms for a [1 200, 425] label matrix, which was most of what
// an item of the training set cost.  Called from the loader threads (ctypes releases the interpreter lock).
extern "C" int itts_normalise_rows_f32(const float* h_x, int64_t rows, int cols, const double* h_sub,
                                       const double* h_div, float* h_out) {
  if (rows < 0 || cols <= 0 || (rows > 0 && (!h_x || !h_out)) || !h_sub || !h_div) {
    itts::set_error("itts_normalise_rows_f32: bad arguments");
    return ITTS_E_INVALID;
  }
  for (int64_t r = 0; r < rows; ++r) {
    const float* x = h_x + r * cols;
    float* o = h_out + r * cols;
    for (int c = 0; c < cols; ++c) o[c] = (float)(((double)x[c] - h_sub[c]) / h_div[c]);
  }
  return ITTS_OK;
}
