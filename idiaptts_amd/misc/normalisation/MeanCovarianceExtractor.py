"""Online mean / covariance statistics with the reference's interface and file formats
(idiaptts/misc/normalisation/MeanCovarianceExtractor.py:20-212); legacy `.bin` layout:
(int32, int32) header + float (size, D).  The sums themselves live in _moments.MomentSums."""
import numpy as np

from ._moments import MomentSums


class MeanCovarianceExtractor(MomentSums):
    file_name_appendix = "mean-covariance"
    second_name = "sum_product_frames"
    param_names = ("mean", "covariance")

    def _first_of(self, block):
        return block.sum(axis=0, keepdims=True)          # the reference keeps a [1, D] row here

    def _second_of(self, block):
        return block.T @ block

    def get_params(self):
        n = self.sum_length
        mean = self.sum_frames / n
        covariance = self.sum_product_frames / n - mean.T @ mean
        return np.atleast_2d(mean, covariance)

    def save_mean_covariance(self, filename, datatype=np.float64):
        self._save_params(filename, datatype)

    @staticmethod
    def load(file_path, datatype=np.float64):
        if file_path.endswith(".bin"):      # legacy: (int32, int32 rows) in front of the (rows, D) block
            rows = int(np.fromfile(file_path, dtype=np.int32, count=2)[1])
            block = np.fromfile(file_path, dtype=datatype, offset=8).reshape((rows, -1))
            mean, covariance = block[:1], block[1:]
        else:
            with np.load(file_path) as archive:
                mean, covariance = archive["mean"], archive["covariance"]
        std_dev = np.sqrt(np.diag(covariance), dtype=np.float32)
        return (mean.squeeze().astype(np.float32, copy=False),
                np.atleast_2d(covariance.astype(np.float32, copy=False)), std_dev.squeeze())
