"""Running minimum / maximum per feature and min-max normalisation (reference
misc/normalisation/MinMaxExtractor.py:20-165): question labels are scaled to [0, 1] with these.
Files: `<prefix->min-max.npz` {min, max}; legacy `.bin` = the two float64 rows back to back."""
import logging
import os

import numpy as np


class MinMaxExtractor(object):
    file_name_appendix = "min-max"

    def __init__(self):
        self.combined_min = float("inf")
        self.combined_max = -float("inf")

    @staticmethod
    def _fix_range_inplace(range_):
        """Zero ranges (constant features) and negative ranges become 1."""
        range_[range_ == 0] = 1
        if (range_ < 0).any():
            logging.warning("Found negative range(s), setting them to 1.")
            range_[range_ < 0] = 1

    def _normalise(self, feature, min_, max_):
        range_ = max_ - min_
        self._fix_range_inplace(range_)
        return (feature - min_) / range_

    def _denormalise(self, feature, min_, max_):
        range_ = max_ - min_
        self._fix_range_inplace(range_)
        return feature * range_ + min_

    def add_sample(self, sample):
        self.combined_min = np.minimum(self.combined_min, sample.min(axis=0))
        self.combined_max = np.maximum(self.combined_max, sample.max(axis=0))

    def get_params(self):
        return self.combined_min, self.combined_max

    def save(self, filename, datatype=np.float64):
        if filename is not None and os.path.basename(filename) != "":
            filename += "-"
        min_, max_ = self.get_params()
        self._save(filename + self.file_name_appendix, {"min": min_, "max": max_}, datatype)

    @staticmethod
    def _save(filename, stats, datatype):
        if datatype is str:
            np.savetxt(filename + ".txt", np.stack(list(stats.values()), axis=0))
        elif datatype in (np.float32, np.float64):
            np.savez(filename, **stats)
        else:
            raise ValueError("Unknown datatype {}".format(datatype))

    @staticmethod
    def load(file_path, datatype=np.float64):
        if datatype is str:
            mm = np.loadtxt(file_path, dtype=np.float32).reshape((2, -1))
            return mm[0], mm[1]
        if file_path.endswith(".bin"):
            mm = np.fromfile(file_path, dtype=datatype).reshape((2, -1))
            return mm[0], mm[1]
        a = np.load(file_path)
        return a["min"].squeeze(), a["max"].squeeze()

    @staticmethod
    def combine_min_max(file_list, dir_out=None, datatype=np.float64, save_txt=False):
        min_, max_ = float("inf"), -float("inf")
        for f in file_list:
            cur_min, cur_max = MinMaxExtractor.load(f, datatype=datatype)
            min_ = np.minimum(min_, cur_min.squeeze())
            max_ = np.maximum(max_, cur_max.squeeze())
        if dir_out is not None:
            filename = os.path.join(dir_out, MinMaxExtractor.file_name_appendix)
            MinMaxExtractor._save(filename, {"min": min_, "max": max_}, datatype)
            if save_txt:
                MinMaxExtractor._save(filename, {"min": min_, "max": max_}, str)
        return min_, max_
