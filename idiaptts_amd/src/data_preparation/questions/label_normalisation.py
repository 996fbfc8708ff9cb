"""HTS full-context labels with state alignment -> frame-level question labels: SURVEY.md section 8(f)
row 2, the producer of the acoustic model's 409 / 425-dim input.  Behaviour of the reference's
HTSLabelNormalisation (data_preparation/questions/label_normalisation.py: question-set loading
:817-897, pattern matching :753-791, load_labels_with_state_alignment :521-666,
perform_normalisation :38-76), rebuilt around whole-utterance numpy arrays:

  * a question file line is `QS "name" {pat,pat,...}` (binary: does any pattern occur in the
    label) or `CQS "name" {pat}` (numeric: the number captured by the pattern, -1 if absent).
    HTK patterns use `*` as wildcard; a pattern that does not start / end with `*` is anchored at
    that end.  All alternatives of one QS become ONE compiled alternation;
  * a label file has five state lines per phone, `start end context[k]` in 100 ns units with
    k = 2..6; a state lasts int((end - start) / 50000) frames of 5 ms;
  * every frame of a phone carries the phone's question vector plus nine sub-phone features
    (fraction through the state forwards / backwards, state length, state index forwards /
    backwards, phone length, state share of the phone, fraction through the phone backwards /
    forwards).  The question vector is matched once per distinct context string (cached across
    utterances), the nine features are computed for all frames of the utterance at once.
Results are bit-identical to the reference's (same IEEE divisions, float64 then float32).

The production path is native: `csrc/labels.cpp` behind the C ABI (itts_questions_load,
itts_labels_count_frames, itts_labels_generate -- patterns compiled once per question, files spread
over plain threads).  The numpy formulation below is kept as its
checker (`HTSLabelNormalisation(..., native=False)`; tests/test_host_logic.py pins both to the
reference's `questions/*.questions` fixtures)."""
import ctypes
import logging
import os
import re
from collections import OrderedDict

import numpy as np

from idiaptts_amd.misc.normalisation.MinMaxExtractor import MinMaxExtractor


def wildcards_to_regex(question, convert_number_pattern=False):
    """HTK question pattern -> regular expression (reference :866-897)."""
    prefix = postfix = ""
    if '*' in question:
        if not question.startswith('*'):
            prefix = r"\A"
        if not question.endswith('*'):
            postfix = r"\Z"
    question = re.escape(question.strip('*')).replace('\\*', '.*')
    question = prefix + question + postfix
    if convert_number_pattern:       # keep the capture groups of CQS patterns alive
        question = question.replace('\\(\\\\d\\+\\)', r'(\d+)')
        question = question.replace('\\(\\[\\\\d\\\\\\.\\]\\+\\)', r'([\d\.]+)')
    return question


class QuestionSet(object):
    """Compiled question file: `binary` holds one alternation per QS line, `continuous` one
    pattern with a capture group per CQS line, both in file order."""

    def __init__(self, file_questions):
        self.binary, self.continuous, self.names = [], [], []
        with open(file_questions) as f:
            for line in f.readlines():
                line = line.replace('\n', '')
                if len(line) <= 5:
                    continue
                patterns = line.split('{')[1].split('}')[0].strip().split(',')
                fields = line.split(' ')
                kind, key = fields[0], fields[1]
                if kind == 'CQS':
                    assert len(patterns) == 1
                    self.continuous.append(re.compile(wildcards_to_regex(patterns[0], True)))
                elif kind == 'QS':
                    anchor = '^' if 'LL-' in key else ''
                    alts = [anchor + wildcards_to_regex(p) for p in patterns]
                    self.binary.append(re.compile("|".join("(?:%s)" % a for a in alts)))
                else:
                    raise ValueError("The question set is not defined correctly: " + line)
                self.names.append(key)
        self._cache = {}

    def __len__(self):
        return len(self.binary) + len(self.continuous)

    def vector(self, label):
        """[QS answers (0/1) ..., CQS values ...] of one context string, float64."""
        v = self._cache.get(label)
        if v is None:
            v = np.empty(len(self), dtype=np.float64)
            nb = len(self.binary)
            for i, q in enumerate(self.binary):
                v[i] = 1.0 if q.search(label) is not None else 0.0
            for i, q in enumerate(self.continuous):
                m = q.search(label)
                v[nb + i] = float(m.group(1)) if m is not None else -1.0
            self._cache[label] = v
        return v


class HTSLabelNormalisation(object):
    logger = logging.getLogger(__name__)
    htk_label_extension = ".lab"
    questions_label_extension = ".questions"
    state_number = 5

    def __init__(self, file_questions=None, add_frame_features=True, subphone_feats='full',
                 continuous_flag=True, native=True, n_threads=None):
        if not add_frame_features or subphone_feats != 'full':
            raise NotImplementedError("Only frame-level labels with the nine 'full' sub-phone "
                                      "features are generated here (what the trainers use).")
        self.file_questions = file_questions
        self.questions = QuestionSet(file_questions)
        self.dict_size = len(self.questions)
        self.frame_feature_size = 9
        self.dimension = self.dict_size + self.frame_feature_size if self.dict_size else 0
        self.n_threads = n_threads or max(1, min(16, (os.cpu_count() or 2) // 2))
        self._handle = None
        if native:
            from idiaptts_amd import lib as _lib
            self._lib = _lib
            L = _lib.load()
            handle, nb, nc = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
            _lib.check(L.itts_questions_load(os.fsencode(file_questions), ctypes.byref(handle),
                                             ctypes.byref(nb), ctypes.byref(nc)),
                       "itts_questions_load")
            assert nb.value + nc.value == self.dict_size
            self._handle = handle

    def __del__(self):
        if getattr(self, "_handle", None):
            self._lib.load().itts_questions_free(self._handle)
            self._handle = None

    def generate_batch(self, file_names):
        """Labels of several `.lab` files in one native call: ([sum frames, dimension] float64,
        frame offsets [n + 1])."""
        L = self._lib.load()
        n = len(file_names)
        paths = (ctypes.c_char_p * n)(*[os.fsencode(f) for f in file_names])
        frames = (ctypes.c_int64 * n)()
        self._lib.check(L.itts_labels_count_frames(paths, n, frames, self.n_threads),
                        "itts_labels_count_frames")
        off = np.concatenate([[0], np.cumsum(np.frombuffer(frames, dtype=np.int64))]).astype(np.int64)
        out = np.empty((int(off[-1]), self.dimension), dtype=np.float64)
        offs = (ctypes.c_int64 * (n + 1))(*[int(o) for o in off])
        self._lib.check(L.itts_labels_generate(self._handle, paths, n, offs, out.ctypes.data,
                                               self.dimension, self.n_threads),
                        "itts_labels_generate")
        return out, off

    # reference method names for the two halves of a question vector
    def pattern_matching_binary(self, label):
        return self.questions.vector(label)[None, :len(self.questions.binary)]

    def pattern_matching_continous_position(self, label):
        return self.questions.vector(label)[None, len(self.questions.binary):]

    @staticmethod
    def parse_state_alignment(file_name):
        """-> (frames per state line, state index 1..5 per line, context string per line)."""
        frames, states, labels = [], [], []
        with open(file_name) as f:
            for line in f.readlines():
                line = line.strip()
                if len(line) < 1:
                    continue
                fields = re.split(r'\s+', line)
                if len(fields) == 1:
                    raise NotImplementedError("Labels without time alignment carry no frames.")
                frames.append(int((int(fields[1]) - int(fields[0])) / 50000))
                full = fields[2]
                states.append(int(full[len(full) - 2]) - 1)      # "...[k]", k = 2..6
                labels.append(full[:len(full) - 3])
        return np.array(frames, dtype=np.int64), np.array(states, dtype=np.int64), labels

    def load_labels_with_state_alignment(self, file_name):
        """[frames, dict_size + 9] float64 (reference :521-666)."""
        if self._handle is not None:
            return self.generate_batch([file_name])[0]
        n, state, labels = self.parse_state_alignment(file_name)
        S = self.state_number
        if len(n) % S != 0 or not np.array_equal(state, np.tile(np.arange(1, S + 1), len(n) // S)):
            raise ValueError("{}: expected {} state lines [2]..[{}] per phone".format(
                file_name, S, S + 1))
        n_phone = n.reshape(-1, S)
        phone_dur = np.repeat(n_phone.sum(axis=1), S)                          # per state line
        base = (np.cumsum(n_phone, axis=1) - n_phone).reshape(-1)              # frames before it
        vectors = np.stack([self.questions.vector(labels[p * S]) for p in range(len(n) // S)]) \
            if len(n) else np.zeros((0, self.dict_size))
        total = int(n.sum())
        line_of = np.repeat(np.arange(len(n)), n)                              # state line per frame
        i = np.arange(total) - np.repeat(np.cumsum(n) - n, n)                  # frame index in state
        nf = n[line_of].astype(np.float64)
        pd = phone_dur[line_of].astype(np.float64)
        bs = base[line_of]
        out = np.empty((total, self.dimension), dtype=np.float64)
        d = self.dict_size
        out[:, :d] = vectors[line_of // S]
        out[:, d] = (i + 1).astype(np.float64) / nf
        out[:, d + 1] = (n[line_of] - i).astype(np.float64) / nf
        out[:, d + 2] = nf
        out[:, d + 3] = state[line_of].astype(np.float64)
        out[:, d + 4] = (6 - state[line_of]).astype(np.float64)
        out[:, d + 5] = pd
        out[:, d + 6] = nf / pd
        out[:, d + 7] = (phone_dur[line_of] - i - bs).astype(np.float64) / pd
        out[:, d + 8] = (bs + i + 1).astype(np.float64) / pd
        return out

    def extract_linguistic_features(self, file_id, out_file_name=None, label_type="state_align",
                                    dur_file_name=None):
        if label_type != "state_align":
            raise NotImplementedError("Only state-aligned labels are supported.")
        labels = self.load_labels_with_state_alignment(file_id)
        if out_file_name:
            labels = np.array(labels, np.float32)
            os.makedirs(os.path.dirname(out_file_name) or ".", exist_ok=True)
            np.savez(os.path.splitext(out_file_name)[0], **{"questions": labels})
        return labels

    def perform_normalisation(self, file_id_list, id_list, dir_labels, dir_out, return_dict=False):
        """Question labels of every id (saved as `<dir_out>/<id>.npz` when dir_out is given) and
        their min / max (`<dir_out>/<id list name>-min-max.npz`); reference :38-76."""
        extractor = MinMaxExtractor()
        dict_labels = OrderedDict()
        if self._handle is not None and len(id_list) > 0:
            # native path, a bounded number of files per call: a chunk's [frames, dim] float64 block
            # (+ its float32 copy for the archives) is all that is alive at a time, so the memory
            # does not grow with the corpus (the reference works file by file); blocks are only
            # kept when the caller asks for the dictionary
            L = self._lib.load()
            chunk = max(1, int(os.environ.get("ITTS_LABEL_CHUNK_FILES", "256")))
            if dir_out is not None:
                os.makedirs(dir_out, exist_ok=True)
            for c0 in range(0, len(id_list), chunk):
                ids = id_list[c0:c0 + chunk]
                files = [os.path.join(dir_labels, i + self.htk_label_extension) for i in ids]
                block, off = self.generate_batch(files)
                if dir_out is not None:
                    block = block.astype(np.float32)          # what extract_linguistic_features saves
                    n = len(ids)
                    paths = (ctypes.c_char_p * n)(*[os.fsencode(os.path.join(dir_out, i + ".npz"))
                                                    for i in ids])
                    for i in ids:
                        os.makedirs(os.path.dirname(os.path.join(dir_out, i)) or ".", exist_ok=True)
                    offs = (ctypes.c_int64 * (n + 1))(*[int(o) for o in off])
                    self._lib.check(L.itts_write_feature_archives(
                        block.ctypes.data, self.dimension, offs, n, paths, 1, (ctypes.c_int * 1)(0),
                        (ctypes.c_int * 1)(self.dimension), (ctypes.c_int * 1)(1),
                        (ctypes.c_char_p * 1)(b"questions"), self.n_threads, None),
                        "itts_write_feature_archives")
                for k, file_id in enumerate(ids):
                    labels = block[off[k]:off[k + 1]]
                    extractor.add_sample(labels)
                    if return_dict:
                        dict_labels[file_id] = labels.copy() if len(id_list) > chunk else labels
                del block
            id_list = []
        for file_id in id_list:
            out = os.path.join(dir_out, file_id + self.questions_label_extension) \
                if dir_out is not None else None
            labels = self.extract_linguistic_features(
                os.path.join(dir_labels, file_id + self.htk_label_extension), out)
            extractor.add_sample(labels)
            if return_dict:
                dict_labels[file_id] = labels
        if dir_out is not None:
            extractor.save(os.path.join(dir_out,
                                        os.path.splitext(os.path.basename(file_id_list))[0]))
        norm_params = extractor.get_params()
        return (dict_labels, norm_params) if return_dict else norm_params
