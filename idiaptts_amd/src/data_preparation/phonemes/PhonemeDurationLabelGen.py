"""Phoneme-duration labels with the reference's exact numeric semantics
(idiaptts/src/data_preparation/phonemes/PhonemeDurationLabelGen.py:175-200, :306-314).
Integer / indexing work: stays on the host, must be bit-exact (north-star)."""
import os

import numpy as np

from idiaptts_amd.misc.normalisation.MeanStdDevExtractor import MeanStdDevExtractor
from idiaptts_amd.src.data_preparation.DataReaders import ReaderBase


class PhonemeDurationLabelGen(ReaderBase):
    dir_labels = "dur"
    ext_durations = ".dur"
    num_states = 5
    min_length = 50000  # one 5 ms frame in HTK 100 ns units

    def __init__(self, dir_labels=None, norm_type="mean_stddev", load_as_matrix=False, **kwargs):
        """Reader over `<dir_labels>/<id>.npz` ('dur') or legacy raw-float32 `.dur` files with
        mean / std-dev normalisation (reference :40-125); `load_as_matrix` returns the hard
        attention matrix of the summed state durations instead (never normalised)."""
        self.directory = dir_labels
        self.load_as_matrix = load_as_matrix
        self.norm_type = None if load_as_matrix else norm_type
        self.norm_params = None
        self._configure("durations")

    def load(self, id_name):
        sample = self.load_sample(id_name, self.directory)
        return self.convert_to_matrix(sample) if self.load_as_matrix else sample

    def get_normalisation_params(self, dir_out=None, file_name=None):
        """(mean, std_dev): `<dir>/[<name>-]mean-std_dev.npz` or the legacy `.bin`."""
        if self.norm_type is None:
            return None
        directory = self.directory if dir_out is None else dir_out
        prefix = "" if file_name is None or os.path.basename(file_name) == "" else file_name + "-"
        base = os.path.join(directory, prefix + MeanStdDevExtractor.file_name_appendix)
        self.norm_params = MeanStdDevExtractor.load(base + (".npz" if os.path.isfile(base + ".npz")
                                                            else ".bin"))
        return self.norm_params

    def preprocess_sample(self, sample):
        if self.norm_type is None:
            return sample
        mean, std_dev = self.norm_params
        return ((sample - mean) / std_dev).astype(np.float32, copy=False)

    def postprocess_sample(self, sample):
        if self.norm_type is None:
            return sample
        mean, std_dev = self.norm_params
        return sample * std_dev + mean

    @staticmethod
    def _get_full_state_align_dur(file_path, min_length: int = 50000, num_states: int = 5):
        """[P, num_states] float32 durations in frames from an HTK state-aligned label file.
        The reference parses the times AS FLOAT32 and divides in float32 (:311), so times above
        2^24 lose integer exactness -- reproduced here on purpose."""
        with open(file_path, 'r') as f:
            htk_labels = [line.rstrip('\n').split()[:2] for line in f]
        timings = np.array(htk_labels, dtype=np.float32) / min_length
        dur = timings[:, 1] - timings[:, 0]
        return dur.reshape(-1, num_states).astype(np.float32)

    @staticmethod
    def durations_to_hard_attention_matrix(durations):
        """durations [P] ints -> selection matrix [sum(durations), P] float32 (reference :175-200)."""
        durations = np.asarray(durations)
        ends = np.cumsum(durations)
        starts = ends - durations
        frames = np.arange(int(ends[-1]) if len(ends) else 0)
        A = ((frames[:, None] >= starts[None, :]) & (frames[:, None] < ends[None, :])) \
            .astype(np.float32)
        assert (A.sum(axis=1) == 1.0).all()
        assert (A.sum(axis=0) == durations).all()
        return A

    @staticmethod
    def convert_to_matrix(sample):
        return PhonemeDurationLabelGen.durations_to_hard_attention_matrix(
            sample.sum(axis=1).astype(int))

    @staticmethod
    def load_sample(id_name, dir_out):
        """`.npz` with key 'dur', or the legacy raw float32 `.dur` file (reference :204-230)."""
        base = os.path.join(dir_out, os.path.basename(id_name))
        if os.path.isfile(base + ".npz"):
            return np.load(base + ".npz")["dur"]
        return np.fromfile(base + PhonemeDurationLabelGen.ext_durations,
                           dtype=np.float32).reshape(-1, PhonemeDurationLabelGen.num_states)

    @staticmethod
    def gen_data(dir_in, dir_out, id_list, label_ext=".lab", return_dict=False):
        norm = MeanStdDevExtractor()
        out = {}
        for name in id_list:
            dur = PhonemeDurationLabelGen._get_full_state_align_dur(
                os.path.join(dir_in, name + label_ext), PhonemeDurationLabelGen.min_length,
                PhonemeDurationLabelGen.num_states)
            if dir_out is not None:
                os.makedirs(dir_out, exist_ok=True)
                np.savez(os.path.join(dir_out, os.path.basename(name) + ".npz"), dur=dur)
            norm.add_sample(dur)
            out[name] = dur
        mean, std = norm.get_params()
        return (out, mean, std) if return_dict else (mean, std)
