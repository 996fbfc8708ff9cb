"""Reader side of the question labels (reference data_preparation/questions/QuestionLabelGen.py:
load with the legacy raw-float32 fallback :119-131, min-max normalisation parameters with the
legacy `.bin` fallback :133-149) and their generation from HTK label files (gen_data :150-203,
on top of questions/label_normalisation.py)."""
import glob
import logging
import os

import numpy as np

from idiaptts_amd import lib as _lib
from idiaptts_amd.src.data_preparation.DataReaders import ReaderBase


class QuestionLabelGen(ReaderBase):
    ext_question = ".questions"
    norm_file_appendix = "min-max"       # MinMaxExtractor.file_name_appendix

    def __init__(self, dir_labels, num_questions=None, features="questions"):
        self.dir_labels = dir_labels
        self.directory = [dir_labels] if isinstance(dir_labels, (str, os.PathLike)) \
            else list(dir_labels)
        self.num_questions = num_questions
        self.features = [features] if isinstance(features, str) else list(features)
        self.norm_params = None
        self._configure("questions")

    @staticmethod
    def load_sample(id_name, dir_out=None, num_questions=None):
        return QuestionLabelGen(dir_out, num_questions).load(id_name)

    def load(self, id_name):
        id_name = os.path.splitext(os.path.basename(id_name))[0]
        for d in self.directory:
            path = os.path.join(d, id_name + ".npz")
            if os.path.isfile(path):
                archive = np.load(path)
                feats = [archive[f].astype(np.float32, copy=False) for f in self.features]
                return feats[0] if len(feats) == 1 else np.concatenate(feats, axis=1)
        labels = np.fromfile(os.path.join(self.directory[0], id_name + self.ext_question),
                             dtype=np.float32)
        return labels.reshape(-1, self.num_questions)

    def get_normalisation_params(self, dir_out=None, file_name=None):
        """(min, max); `<dir>/<prefix->min-max.npz` or the legacy float64 `.bin` holding the two
        rows back to back (MinMaxExtractor.load :100-122)."""
        directory = self.directory[0] if dir_out is None else dir_out
        prefix = "" if file_name is None or os.path.basename(file_name) == "" \
            else file_name + "-"
        base = os.path.join(directory, prefix + self.norm_file_appendix)
        if os.path.isfile(base + ".npz"):
            a = np.load(base + ".npz")
            self.norm_params = (a["min"].squeeze(), a["max"].squeeze())
        else:
            mm = np.fromfile(base + ".bin", dtype=np.float64).reshape((2, -1))
            self.norm_params = (mm[0], mm[1])
        return self.norm_params

    @staticmethod
    def _range(min_, max_):
        rng = np.array(max_ - min_)
        rng[rng == 0] = 1                # MinMaxExtractor._fix_range_inplace
        return rng

    def preprocess_sample(self, sample, norm_params=None):
        min_, max_ = self.norm_params if norm_params is None else norm_params
        return _lib.normalise_rows(sample, min_, self._range(min_, max_))

    def postprocess_sample(self, sample, norm_params=None):
        min_, max_ = self.norm_params if norm_params is None else norm_params
        return sample * self._range(min_, max_) + min_

    # ------------------------------------------------------------------------------ generation
    @staticmethod
    def gen_data(dir_in, file_questions, dir_out=None, file_id_list="", id_list=None,
                 return_dict=False):
        """Question labels from HTK full-context labels with state alignment (`<dir_in>/<id>.lab`):
        one `<dir_out>/<id>.npz` {"questions": float32 [frames, n_questions + 9]} per id and the
        min / max over all of them in `<dir_out>/<id list name>-min-max.npz`.  Returns
        (min, max), preceded by an OrderedDict id -> labels when return_dict."""
        from idiaptts_amd.src.data_preparation.questions.label_normalisation import \
            HTSLabelNormalisation
        if id_list is None:
            id_list = [os.path.splitext(os.path.basename(f))[0]
                       for f in glob.glob(os.path.join(dir_in, "*.lab"))]
            file_id_list_name = "all"
        else:
            file_id_list_name = os.path.splitext(os.path.basename(file_id_list))[0]
            id_list = ['{}'.format(os.path.basename(e)) for e in id_list]
        if dir_out is not None:
            os.makedirs(dir_out, exist_ok=True)
        operator = HTSLabelNormalisation(file_questions)
        out = operator.perform_normalisation(file_id_list_name, id_list, dir_in, dir_out,
                                             return_dict=return_dict)
        if return_dict:
            return out[0], out[1][0], out[1][1]
        return out[0], out[1]

    @staticmethod
    def questions_to_phoneme_indices(questions, np_phoneme_indices_in_question, default_index=-1):
        """Index (within the given phoneme questions) of the phoneme of every frame; frames without
        any phoneme get `default_index` (reference :216-241)."""
        sub = questions[:, np_phoneme_indices_in_question]
        indices = sub.argmax(axis=1)
        no_phoneme = sub.max(axis=1) == 0
        if no_phoneme.any():
            logging.warning("Using default phoneme index {} for frames without phoneme "
                            "information which are {}".format(default_index,
                                                              np.flatnonzero(no_phoneme)))
        indices[no_phoneme] = default_index
        return indices

    @staticmethod
    def questions_to_phoneme_per_frame(questions, np_phoneme_indices_in_question_file,
                                       question_file):
        """Phoneme name of every frame, '?' where none is set (reference :243-278)."""
        with open(question_file) as f:
            lines = np.array(f.readlines())[np_phoneme_indices_in_question_file]
        names = [l.split()[1].replace('C_', '').replace('C-', '').replace('\"', '') for l in lines]
        idx = QuestionLabelGen.questions_to_phoneme_indices(
            questions, np_phoneme_indices_in_question_file) + 1
        return np.insert(np.array(names), 0, '?', axis=0)[idx]

    @staticmethod
    def questions_to_phonemes(questions, np_phoneme_indices_in_question_file, question_file):
        """(first frame, phoneme) for every run of equal phonemes (reference :280-309)."""
        per_frame = QuestionLabelGen.questions_to_phoneme_per_frame(
            questions, np_phoneme_indices_in_question_file, question_file)
        phonemes = [(0, per_frame[0])]
        for index, phoneme in enumerate(per_frame):
            if phonemes[-1][1] != phoneme:
                phonemes.append((index, phoneme))
        return np.array(phonemes)
