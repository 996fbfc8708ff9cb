"""Online mean / std-dev statistics with the reference's interface and file formats
(idiaptts/misc/normalisation/MeanStdDevExtractor.py:20-204): `.npz` archives with keys
{sum_frames, sum_squared_frames | mean, std_dev, sum_length} and the legacy `.bin` layout
(int32 sum_length + float (2, D))."""
import os
import struct

import numpy as np


class MeanStdDevExtractor(object):
    file_name_stats = "stats"
    file_name_appendix = "mean-std_dev"

    def __init__(self):
        self.sum_length = 0
        self.sum_frames = 0
        self.sum_squared_frames = 0

    def _normalise(self, feature, mean, std_dev):
        return (feature - mean) / std_dev

    def _denormalise(self, feature, mean, std_dev):
        return feature * std_dev + mean

    def add_sample(self, sample):
        assert sample is not None, "Sample cannot be None."
        self.sum_length += len(sample)
        self.sum_frames += np.sum(sample, axis=0)
        self.sum_squared_frames += np.sum(sample**2, axis=0)

    def get_params(self):
        mean = self.sum_frames / self.sum_length
        std_dev = np.sqrt(self.sum_squared_frames / self.sum_length - mean**2)
        return np.atleast_1d(mean), np.atleast_1d(std_dev)

    def combine(self, other):
        """Statistics are additive (what reference combine_stats :163-204 and the multi-GPU
        all-reduce of SURVEY.md section 8e rely on)."""
        self.sum_length += other.sum_length
        self.sum_frames = self.sum_frames + other.sum_frames
        self.sum_squared_frames = self.sum_squared_frames + other.sum_squared_frames

    def save(self, filename, datatype=np.float64):
        self.save_stats(filename, datatype)
        self.save_mean_std_dev(filename, datatype)

    @staticmethod
    def _prefix(filename):
        if filename is not None and os.path.basename(filename) != "":
            filename += "-"
        return filename

    def save_stats(self, filename, datatype=np.float64):
        self._save(self._prefix(filename) + self.file_name_stats, self.sum_length,
                   {"sum_frames": self.sum_frames,
                    "sum_squared_frames": self.sum_squared_frames}, datatype)

    def save_mean_std_dev(self, filename, datatype=np.float64):
        mean, std_dev = self.get_params()
        self._save(self._prefix(filename) + self.file_name_appendix, self.sum_length,
                   {"mean": mean, "std_dev": std_dev}, datatype)

    @staticmethod
    def _save(filename, sum_length, stats, datatype):
        if datatype is str:
            np.savetxt(filename + ".txt", np.concatenate(list(stats.values()), axis=0),
                       header=str(sum_length))
        elif datatype is np.float32 or datatype is np.float64:
            stats = {k: np.atleast_1d(v).astype(datatype, copy=False) for k, v in stats.items()}
            stats["sum_length"] = np.array(sum_length, dtype=int)
            np.savez(filename, **stats)
        else:
            raise ValueError("Unknown datatype {}".format(datatype))

    @staticmethod
    def load_stats(file_path, datatype=np.float64):
        a = np.load(file_path)
        return a["sum_frames"], a["sum_squared_frames"], a["sum_length"]

    @staticmethod
    def load(file_path, datatype=np.float64):
        if file_path.endswith(".bin"):  # legacy
            with open(file_path, 'rb') as f:
                _ = struct.unpack("i", f.read(4))[0]
                mean_std_dev = np.fromfile(f, dtype=datatype).reshape((2, -1))
            mean, std_dev = np.split(mean_std_dev, mean_std_dev.shape[0], axis=0)
        else:
            a = np.load(file_path)
            mean, std_dev = a["mean"], a["std_dev"]
        return (mean.astype(np.float32, copy=False), std_dev.astype(np.float32, copy=False))
