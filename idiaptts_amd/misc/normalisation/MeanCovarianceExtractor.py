"""Online mean / covariance statistics with the reference's interface and file formats
(idiaptts/misc/normalisation/MeanCovarianceExtractor.py:20-212); legacy `.bin` layout:
(int32, int32) header + float (size, D)."""
import os
import struct

import numpy as np


class MeanCovarianceExtractor(object):
    file_name_stats = "stats"
    file_name_appendix = "mean-covariance"

    def __init__(self):
        self.sum_length = 0
        self.sum_frames = 0
        self.sum_product_frames = 0

    def _normalise(self, feature, mean, std_dev):
        return (feature - mean) / std_dev

    def _denormalise(self, feature, mean, std_dev):
        return feature * std_dev + mean

    def add_sample(self, sample):
        assert sample is not None, "Sample cannot be None."
        self.sum_length += len(sample)
        self.sum_frames += np.sum(sample, axis=0, keepdims=True)
        self.sum_product_frames += np.dot(np.transpose(sample), sample)

    def get_params(self):
        mean = self.sum_frames / self.sum_length
        mean_product = np.dot(np.transpose(mean), mean)
        covariance = self.sum_product_frames / self.sum_length - mean_product
        return np.atleast_2d(mean, covariance)

    def combine(self, other):
        self.sum_length += other.sum_length
        self.sum_frames = self.sum_frames + other.sum_frames
        self.sum_product_frames = self.sum_product_frames + other.sum_product_frames

    def save(self, filename, datatype=np.float64):
        self.save_stats(filename, datatype)
        self.save_mean_covariance(filename, datatype)

    @staticmethod
    def _prefix(filename):
        if filename is not None and os.path.basename(filename) != "":
            filename += "-"
        return filename

    def save_stats(self, filename, datatype=np.float64):
        self._save(self._prefix(filename) + self.file_name_stats, self.sum_length,
                   {"sum_frames": self.sum_frames,
                    "sum_product_frames": self.sum_product_frames}, datatype)

    def save_mean_covariance(self, filename, datatype=np.float64):
        mean, covariance = self.get_params()
        self._save(self._prefix(filename) + self.file_name_appendix, self.sum_length,
                   {"mean": mean, "covariance": covariance}, datatype)

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
        return a['sum_frames'], a['sum_product_frames'], a['sum_length']

    @staticmethod
    def load(file_path, datatype=np.float64):
        if file_path.endswith(".bin"):  # legacy
            with open(file_path, 'rb') as f:
                header = struct.unpack("ii", f.read(8))
                size = header[1]
                mean_covariance = np.fromfile(f, dtype=datatype).reshape((size, -1))
                mean, covariance = np.split(mean_covariance, (1,), axis=0)
        else:
            a = np.load(file_path)
            mean, covariance = a["mean"], a['covariance']
        std_dev = np.sqrt(np.diag(covariance), dtype=np.float32)
        return (mean.squeeze().astype(np.float32, copy=False),
                np.atleast_2d(covariance.astype(np.float32, copy=False)), std_dev.squeeze())
