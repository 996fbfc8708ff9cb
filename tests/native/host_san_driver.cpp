// Test driver for the threaded host code of the library (csrc/labels.cpp, csrc/hostio.cpp): built
// host-only with -fsanitize=address,undefined or -fsanitize=thread by tests/test_sanitizers.py and
// run on the fixture files with several worker threads.  It writes what the entry points return
// to <out dir>/labels.f64, wav.f64 and the archives; the test compares them with the production
// library's results.  Not product code.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/idiaptts_amd.h"

namespace itts {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace itts
extern "C" const char* itts_last_error(void) { return itts::g_err.c_str(); }

#define CHECK(x)                                                                       \
  do {                                                                                 \
    int rc_ = (x);                                                                     \
    if (rc_ != 0) {                                                                    \
      fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, itts_last_error());             \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

static std::vector<std::string> read_list(const char* path) {
  std::vector<std::string> v;
  FILE* f = fopen(path, "r");
  if (!f) return v;
  char buf[4096];
  while (fgets(buf, sizeof buf, f)) {
    size_t n = strlen(buf);
    while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;
    if (n) v.push_back(buf);
  }
  fclose(f);
  return v;
}
static int dump(const std::string& path, const void* p, size_t bytes) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return 1;
  const size_t w = fwrite(p, 1, bytes, f);
  fclose(f);
  return w == bytes ? 0 : 1;
}

// usage: driver <question file> <list of .lab files> <list of .wav files> <out dir> <threads>
int main(int argc, char** argv) {
  if (argc != 6) return 2;
  const std::vector<std::string> labs = read_list(argv[2]), wavs = read_list(argv[3]);
  const std::string out = argv[4];
  const int nt = atoi(argv[5]);

  // question labels
  void* qs = nullptr;
  int nb = 0, nc = 0;
  CHECK(itts_questions_load(argv[1], &qs, &nb, &nc));
  const int dim = nb + nc + 9;
  std::vector<const char*> lp;
  for (auto& s : labs) lp.push_back(s.c_str());
  std::vector<int64_t> frames(labs.size()), off(labs.size() + 1, 0);
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(itts_labels_count_frames(lp.data(), (int)lp.size(), frames.data(), nt));
    for (size_t i = 0; i < labs.size(); ++i) off[i + 1] = off[i] + frames[i];
    std::vector<double> block((size_t)off.back() * dim);
    CHECK(itts_labels_generate(qs, lp.data(), (int)lp.size(), off.data(), block.data(), dim, nt));
    if (rep == 2) {
      if (dump(out + "/labels.f64", block.data(), block.size() * 8)) return 3;
      if (dump(out + "/labels.off", off.data(), off.size() * 8)) return 3;
    }
  }
  // errors must come back as errors: a file that does not exist
  {
    const char* bad[1] = {"/nonexistent/x.lab"};
    int64_t fr[1];
    if (itts_labels_count_frames(bad, 1, fr, nt) == 0) return 4;
  }
  double vec[4096];
  CHECK(itts_questions_vector(qs, "x^x-sil+dh=ax@1_0/A:0_0_0/B:0-0-0@1-0&1-0#1-0$1-0!0-0;0-0|0/C:1+1+2", vec));
  itts_questions_free(qs);

  // wav reader
  std::vector<const char*> wp;
  for (auto& s : wavs) wp.push_back(s.c_str());
  std::vector<int64_t> woff(wavs.size() + 1, 0);
  for (size_t i = 0; i < wavs.size(); ++i) {
    int fs = 0;
    int64_t n = 0;
    CHECK(itts_wav_info(wp[i], &fs, &n));
    woff[i + 1] = woff[i] + n;
  }
  std::vector<double> audio((size_t)woff.back());
  for (int rep = 0; rep < 3; ++rep)
    CHECK(itts_wav_read_batch(wp.data(), (int)wp.size(), woff.data(), 0.97, audio.data(), nt));
  if (dump(out + "/wav.f64", audio.data(), audio.size() * 8)) return 3;
  if (dump(out + "/wav.off", woff.data(), woff.size() * 8)) return 3;

  // archive writer: every utterance gets two archives (a 3-part stream and a 1-part stream)
  const int n_utts = (int)wavs.size(), ld = 12;
  std::vector<int64_t> foff(n_utts + 1, 0);
  for (int u = 0; u < n_utts; ++u) foff[u + 1] = foff[u] + 50 + 7 * u;
  std::vector<float> feat((size_t)foff.back() * ld);
  for (size_t i = 0; i < feat.size(); ++i) feat[i] = (float)((i * 2654435761u) % 1000) * 0.001f - 0.5f;
  std::vector<std::string> paths;
  for (int u = 0; u < n_utts; ++u) {
    paths.push_back(out + "/a" + std::to_string(u) + ".npz");
    paths.push_back(out + "/b" + std::to_string(u) + ".npz");
  }
  std::vector<const char*> pp;
  for (auto& s : paths) pp.push_back(s.c_str());
  const int col0[2] = {0, 9}, width[2] = {3, 3}, parts[2] = {3, 1};
  const char* keys[2] = {"cmp_mcep3", "bap"};
  std::vector<unsigned char> merge(paths.size(), 0);
  for (int rep = 0; rep < 2; ++rep)
    CHECK(itts_write_feature_archives(feat.data(), ld, foff.data(), n_utts, pp.data(), 2, col0, width, parts, keys,
                                      nt, rep ? merge.data() : nullptr));
  if (dump(out + "/feat.f32", feat.data(), feat.size() * 4)) return 3;
  if (dump(out + "/feat.off", foff.data(), foff.size() * 8)) return 3;
  printf("ok %d label files, %d wav files, dim %d\n", (int)labs.size(), (int)wavs.size(), dim);
  return 0;
}
