"""First and second moments of a feature stream, kept as additive sums so that shards merge by
addition (one all-reduce under data parallelism) and so that the device path
(csrc/features.hip: itts_feature_stats, the sums of a whole batch in one launch) can hand its sums
over directly.  The two public extractors only differ in what the second moment is (element-wise
squares or the full outer product), in the names of their attributes and files (the reference's
API: idiaptts/misc/normalisation/MeanStdDevExtractor.py, MeanCovarianceExtractor.py) and in the
legacy `.bin` layout they still read."""
import os

import numpy as np


class MomentSums(object):
    """`count`, `first` and `second` live under the reference's attribute names on the subclass
    (`sum_length`, `sum_frames`, `second_name`)."""

    file_name_stats = "stats"
    file_name_appendix = None      # subclass: suffix of the parameter file
    second_name = None             # subclass: attribute / archive key of the second-moment sum
    param_names = None             # subclass: archive keys of the two parameters

    def __init__(self):
        self.sum_length = 0
        self.sum_frames = 0
        setattr(self, self.second_name, 0)

    # -- accumulation -------------------------------------------------------------------------------
    def _second_of(self, block):
        raise NotImplementedError

    def _first_of(self, block):
        return block.sum(axis=0)

    def add_sums(self, count, first, second):
        """Adds sums computed elsewhere (another shard, or the device: world.py feeds the output of
        itts_feature_stats through here)."""
        self.sum_length += count
        self.sum_frames = self.sum_frames + first
        setattr(self, self.second_name, getattr(self, self.second_name) + second)

    def add_sample(self, sample):
        assert sample is not None, "Sample cannot be None."
        block = np.asarray(sample)
        self.add_sums(len(block), self._first_of(block), self._second_of(block))

    def combine(self, other):
        self.add_sums(other.sum_length, other.sum_frames, getattr(other, self.second_name))

    # -- the affine map both directions (the reference keeps them as methods) ------------------------
    @staticmethod
    def _normalise(feature, mean, std_dev):
        return (feature - mean) / std_dev

    @staticmethod
    def _denormalise(feature, mean, std_dev):
        return feature * std_dev + mean

    # -- files ----------------------------------------------------------------------------------------
    @staticmethod
    def _with_suffix(filename, suffix):
        named = filename is not None and os.path.basename(filename) != ""
        return (filename + "-" if named else filename) + suffix

    @staticmethod
    def _write(path, count, arrays, datatype):
        if datatype is str:
            np.savetxt(path + ".txt", np.concatenate(list(arrays.values()), axis=0), header=str(count))
            return
        if datatype not in (np.float32, np.float64):
            raise ValueError("Unknown datatype {}".format(datatype))
        out = {key: np.atleast_1d(val).astype(datatype, copy=False) for key, val in arrays.items()}
        np.savez(path, sum_length=np.array(count, dtype=int), **out)

    def save_stats(self, filename, datatype=np.float64):
        self._write(self._with_suffix(filename, self.file_name_stats), self.sum_length,
                    {"sum_frames": self.sum_frames, self.second_name: getattr(self, self.second_name)},
                    datatype)

    def _save_params(self, filename, datatype):
        self._write(self._with_suffix(filename, self.file_name_appendix), self.sum_length,
                    dict(zip(self.param_names, self.get_params())), datatype)

    def save(self, filename, datatype=np.float64):
        self.save_stats(filename, datatype)
        self._save_params(filename, datatype)

    @classmethod
    def load_stats(cls, file_path, datatype=np.float64):
        with np.load(file_path) as archive:
            return archive["sum_frames"], archive[cls.second_name], archive["sum_length"]
