// Question-label generation on the host (no GPU): HTS full-context labels with state alignment ->
// frame-level question vectors + nine sub-phone features, the acoustic model's 409 / 425-dim input.
// Reference: HTSLabelNormalisation (idiaptts/src/data_preparation/questions/label_normalisation.py:
// question-set loading :817-897, pattern matching :753-791, load_labels_with_state_alignment
// :521-666), SURVEY.md section 8(f) row 2 ("vectorised bit-exact C++, patterns compiled once per
// question").  Results are bit-identical to the reference's: the same IEEE divisions in float64.
//
// A question file line is `QS "name" {pat,pat,...}` (1 if any pattern occurs in the context string)
// or `CQS "name" {pat}` (the number the pattern's group captures, -1 if it does not occur).  The
// reference turns HTK patterns into regular expressions (`*` -> `.*`, everything else literal, a
// pattern that contains `*` but does not start / end with it is anchored at that end, `(\d+)` and
// `([\d\.]+)` in CQS patterns stay capture groups) and calls re.search.  Those expressions only use
// literals, `.*` and one digit group, so they are compiled here into token lists and run by a small
// backtracking matcher with re.search's leftmost / greedy semantics.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/idiaptts_amd.h"

namespace itts {
void set_error(const std::string& msg);
}

namespace {

enum TokKind { LIT, STAR, GROUP_D, GROUP_DD };
struct Tok {
  TokKind kind;
  std::string lit;
};
struct Pattern {
  std::vector<Tok> toks;
  bool anchor_start = false, anchor_end = false;
};

bool is_digit(char c) { return c >= '0' && c <= '9'; }

// match toks[ti..] against s[pos..]; on success *cap = (begin, end) of the group (if any)
bool match_here(const Pattern& p, size_t ti, const std::string& s, size_t pos, int* cb, int* ce) {
  if (ti == p.toks.size()) return !p.anchor_end || pos == s.size();
  const Tok& t = p.toks[ti];
  switch (t.kind) {
    case LIT:
      if (s.compare(pos, t.lit.size(), t.lit) != 0) return false;
      return match_here(p, ti + 1, s, pos + t.lit.size(), cb, ce);
    case STAR:
      for (size_t e = s.size() + 1; e-- > pos;)          // greedy: longest first
        if (match_here(p, ti + 1, s, e, cb, ce)) return true;
      return false;
    default: {
      size_t e = pos;
      while (e < s.size() && (is_digit(s[e]) || (t.kind == GROUP_DD && s[e] == '.'))) ++e;
      for (; e > pos; --e) {                             // one or more, greedy
        if (match_here(p, ti + 1, s, e, cb, ce)) {
          *cb = (int)pos;
          *ce = (int)e;
          return true;
        }
      }
      return false;
    }
  }
}

bool search(const Pattern& p, const std::string& s, int* cb, int* ce) {
  *cb = *ce = -1;
  if (p.toks.size() == 1 && p.toks[0].kind == LIT) {
    // the common case (`*-a+*`, `aa~*`, `*|1`, `abc`): one literal, anchored or not
    const std::string& lit = p.toks[0].lit;
    if (lit.size() > s.size()) return false;
    if (p.anchor_start && p.anchor_end) return s == lit;
    if (p.anchor_start) return s.compare(0, lit.size(), lit) == 0;
    if (p.anchor_end) return s.compare(s.size() - lit.size(), lit.size(), lit) == 0;
    return s.find(lit) != std::string::npos;
  }
  if (p.anchor_start) return match_here(p, 0, s, 0, cb, ce);
  for (size_t st = 0; st <= s.size(); ++st)
    if (match_here(p, 0, s, st, cb, ce)) return true;
  return false;
}

// reference wildcards_to_regex (:866-897) as a token list
Pattern compile(const std::string& q, bool continuous, bool force_start) {
  Pattern p;
  const bool has_star = q.find('*') != std::string::npos;
  if (has_star) {
    p.anchor_start = q.front() != '*';
    p.anchor_end = q.back() != '*';
  }
  if (force_start) p.anchor_start = true;
  size_t a = 0, b = q.size();
  while (a < b && q[a] == '*') ++a;          // str.strip('*')
  while (b > a && q[b - 1] == '*') --b;
  std::string lit;
  auto flush = [&]() {
    if (!lit.empty()) {
      p.toks.push_back({LIT, lit});
      lit.clear();
    }
  };
  for (size_t i = a; i < b;) {
    if (q[i] == '*') {
      flush();
      if (p.toks.empty() || p.toks.back().kind != STAR) p.toks.push_back({STAR, ""});
      ++i;
    } else if (continuous && q.compare(i, 5, "(\\d+)") == 0) {
      flush();
      p.toks.push_back({GROUP_D, ""});
      i += 5;
    } else if (continuous && q.compare(i, 9, "([\\d\\.]+)") == 0) {
      flush();
      p.toks.push_back({GROUP_DD, ""});
      i += 9;
    } else {
      lit.push_back(q[i++]);
    }
  }
  flush();
  return p;
}

struct QuestionSet {
  std::vector<std::vector<Pattern>> binary;   // alternatives of every QS line
  std::vector<Pattern> continuous;            // one pattern per CQS line
  int size() const { return (int)(binary.size() + continuous.size()); }
  void vector_of(const std::string& label, double* v) const {
    int cb, ce;
    for (size_t i = 0; i < binary.size(); ++i) {
      double hit = 0.0;
      for (const Pattern& p : binary[i])
        if (search(p, label, &cb, &ce)) { hit = 1.0; break; }
      v[i] = hit;
    }
    for (size_t i = 0; i < continuous.size(); ++i) {
      double val = -1.0;
      if (search(continuous[i], label, &cb, &ce) && cb >= 0)
        val = strtod(label.substr(cb, ce - cb).c_str(), nullptr);
      v[binary.size() + i] = val;
    }
  }
};

std::vector<std::string> split(const std::string& s, char c) {
  std::vector<std::string> out;
  size_t a = 0;
  for (;;) {
    const size_t b = s.find(c, a);
    if (b == std::string::npos) { out.push_back(s.substr(a)); break; }
    out.push_back(s.substr(a, b - a));
    a = b + 1;
  }
  return out;
}

std::string strip(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) ++a;
  while (b > a && isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}

bool read_lines(const char* path, std::vector<std::string>* lines) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  std::string cur;
  int ch;
  while ((ch = fgetc(f)) != EOF) {
    if (ch == '\n') { lines->push_back(cur); cur.clear(); }
    else cur.push_back((char)ch);
  }
  if (!cur.empty()) lines->push_back(cur);
  fclose(f);
  return true;
}

struct StateLine {
  int64_t frames;
  int state;            // 1..5
  std::string label;
};

// reference parse of a state-aligned label file (:521-560): `start end context[k]`, k = 2..6
bool parse_lab(const char* path, std::vector<StateLine>* out, std::string* err) {
  std::vector<std::string> lines;
  if (!read_lines(path, &lines)) { *err = std::string("cannot open ") + path; return false; }
  for (const std::string& raw : lines) {
    const std::string line = strip(raw);
    if (line.empty()) continue;
    std::vector<std::string> f;
    size_t i = 0;
    while (i < line.size()) {                      // re.split(r'\s+', line)
      size_t j = i;
      while (j < line.size() && !isspace((unsigned char)line[j])) ++j;
      f.push_back(line.substr(i, j - i));
      while (j < line.size() && isspace((unsigned char)line[j])) ++j;
      i = j;
    }
    if (f.size() < 3) { *err = std::string("labels without time alignment carry no frames: ") + path; return false; }
    const long long st = atoll(f[0].c_str()), en = atoll(f[1].c_str());
    const std::string& full = f[2];
    if (full.size() < 3) { *err = std::string("malformed label in ") + path; return false; }
    StateLine sl;
    sl.frames = (int64_t)((double)(en - st) / 50000.0);   // int((end - start) / 50000)
    sl.state = (full[full.size() - 2] - '0') - 1;
    sl.label = full.substr(0, full.size() - 3);
    out->push_back(sl);
  }
  const size_t S = 5;
  if (out->size() % S != 0) { *err = std::string("expected 5 state lines [2]..[6] per phone: ") + path; return false; }
  for (size_t i = 0; i < out->size(); ++i)
    if ((*out)[i].state != (int)(i % S) + 1) { *err = std::string("expected 5 state lines [2]..[6] per phone: ") + path; return false; }
  return true;
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
      if (!job(i, t, &errs[t])) failed.store(true);
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

}  // namespace

extern "C" int itts_questions_load(const char* h_path, void** handle, int* n_binary, int* n_continuous) {
  if (!h_path || !handle) { itts::set_error("itts_questions_load: null pointer"); return ITTS_E_INVALID; }
  std::vector<std::string> lines;
  if (!read_lines(h_path, &lines)) {
    itts::set_error(std::string("itts_questions_load: cannot open ") + h_path);
    return ITTS_E_INVALID;
  }
  QuestionSet* qs = new QuestionSet();
  for (const std::string& line : lines) {
    if (line.size() <= 5) continue;
    const size_t lb = line.find('{');
    if (lb == std::string::npos) { delete qs; itts::set_error("itts_questions_load: line without {...}: " + line); return ITTS_E_INVALID; }
    std::string inner = line.substr(lb + 1);
    const size_t rb = inner.find('}');
    if (rb != std::string::npos) inner = inner.substr(0, rb);
    const std::vector<std::string> pats = split(strip(inner), ',');
    const std::vector<std::string> fields = split(line, ' ');
    const std::string kind = fields[0], key = fields.size() > 1 ? fields[1] : "";
    if (kind == "CQS") {
      if (pats.size() != 1) { delete qs; itts::set_error("itts_questions_load: CQS with several patterns: " + line); return ITTS_E_INVALID; }
      qs->continuous.push_back(compile(pats[0], true, false));
    } else if (kind == "QS") {
      const bool ll = key.find("LL-") != std::string::npos;
      std::vector<Pattern> alts;
      for (const std::string& p : pats) alts.push_back(compile(p, false, ll));
      qs->binary.push_back(alts);
    } else {
      delete qs;
      itts::set_error("itts_questions_load: the question set is not defined correctly: " + line);
      return ITTS_E_INVALID;
    }
  }
  *handle = qs;
  if (n_binary) *n_binary = (int)qs->binary.size();
  if (n_continuous) *n_continuous = (int)qs->continuous.size();
  return ITTS_OK;
}

extern "C" void itts_questions_free(void* handle) { delete static_cast<QuestionSet*>(handle); }

extern "C" int itts_questions_vector(void* handle, const char* h_label, double* h_out) {
  if (!handle || !h_label || !h_out) { itts::set_error("itts_questions_vector: null pointer"); return ITTS_E_INVALID; }
  static_cast<QuestionSet*>(handle)->vector_of(h_label, h_out);
  return ITTS_OK;
}

extern "C" int itts_labels_count_frames(const char* const* h_paths, int n_files, int64_t* h_frames,
                                        int n_threads) {
  if (n_files < 0 || (n_files > 0 && (!h_paths || !h_frames))) {
    itts::set_error("itts_labels_count_frames: null pointer");
    return ITTS_E_INVALID;
  }
  std::string err;
  const bool ok = run_parallel(n_files, n_threads, [&](int i, int, std::string* e) {
    std::vector<StateLine> sl;
    if (!parse_lab(h_paths[i], &sl, e)) return false;
    int64_t t = 0;
    for (const StateLine& s : sl) t += s.frames;
    h_frames[i] = t;
    return true;
  }, &err);
  if (!ok) { itts::set_error("itts_labels_count_frames: " + err); return ITTS_E_INVALID; }
  return ITTS_OK;
}

extern "C" int itts_labels_generate(void* handle, const char* const* h_paths, int n_files,
                                    const int64_t* h_frame_off, double* h_out, int64_t ld_out,
                                    int n_threads) {
  if (!handle || n_files < 0 || (n_files > 0 && (!h_paths || !h_frame_off || !h_out))) {
    itts::set_error("itts_labels_generate: null pointer");
    return ITTS_E_INVALID;
  }
  const QuestionSet* qs = static_cast<QuestionSet*>(handle);
  const int d = qs->size();
  if (ld_out < d + 9) { itts::set_error("itts_labels_generate: ld_out smaller than the label width"); return ITTS_E_INVALID; }
  if (n_threads < 1) n_threads = 1;
  // question vectors are matched once per distinct context string (per worker thread)
  std::vector<std::unordered_map<std::string, std::vector<double>>> cache(n_threads);
  std::string err;
  const bool ok = run_parallel(n_files, n_threads, [&](int i, int t, std::string* e) {
    std::vector<StateLine> sl;
    if (!parse_lab(h_paths[i], &sl, e)) return false;
    int64_t total = 0;
    for (const StateLine& s : sl) total += s.frames;
    if (total != h_frame_off[i + 1] - h_frame_off[i]) { *e = std::string("frame count changed between the two passes: ") + h_paths[i]; return false; }
    double* out = h_out + h_frame_off[i] * ld_out;
    const int S = 5;
    for (size_t ph = 0; ph * S < sl.size(); ++ph) {
      const std::string& label = sl[ph * S].label;
      auto it = cache[t].find(label);
      if (it == cache[t].end()) {
        std::vector<double> v(d);
        qs->vector_of(label, v.data());
        it = cache[t].emplace(label, std::move(v)).first;
      }
      const std::vector<double>& qv = it->second;
      int64_t phone_dur = 0;
      for (int k = 0; k < S; ++k) phone_dur += sl[ph * S + k].frames;
      int64_t base = 0;                        // frames of the phone before this state
      for (int k = 0; k < S; ++k) {
        const StateLine& s = sl[ph * S + k];
        const int64_t n = s.frames;
        const double nf = (double)n, pd = (double)phone_dur;
        for (int64_t f = 0; f < n; ++f, out += ld_out) {
          memcpy(out, qv.data(), sizeof(double) * d);
          out[d] = (double)(f + 1) / nf;
          out[d + 1] = (double)(n - f) / nf;
          out[d + 2] = nf;
          out[d + 3] = (double)s.state;
          out[d + 4] = (double)(6 - s.state);
          out[d + 5] = pd;
          out[d + 6] = nf / pd;
          out[d + 7] = (double)(phone_dur - f - base) / pd;
          out[d + 8] = (double)(base + f + 1) / pd;
        }
        base += n;
      }
    }
    return true;
  }, &err);
  if (!ok) { itts::set_error("itts_labels_generate: " + err); return ITTS_E_INVALID; }
  return ITTS_OK;
}
