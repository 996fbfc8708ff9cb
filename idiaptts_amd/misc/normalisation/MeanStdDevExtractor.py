"""Online mean / std-dev statistics with the reference's interface and file formats
(idiaptts/misc/normalisation/MeanStdDevExtractor.py:20-204): `.npz` archives with keys
{sum_frames, sum_squared_frames | mean, std_dev, sum_length} and the legacy `.bin` layout
(int32 sum_length + float (2, D)).  The sums themselves live in _moments.MomentSums."""
import numpy as np

from ._moments import MomentSums


class MeanStdDevExtractor(MomentSums):
    file_name_appendix = "mean-std_dev"
    second_name = "sum_squared_frames"
    param_names = ("mean", "std_dev")

    def _second_of(self, block):
        return np.square(block).sum(axis=0)

    def get_params(self):
        n = self.sum_length
        mean = self.sum_frames / n
        variance = self.sum_squared_frames / n - mean**2
        return np.atleast_1d(mean), np.atleast_1d(np.sqrt(variance))

    def save_mean_std_dev(self, filename, datatype=np.float64):
        self._save_params(filename, datatype)

    @staticmethod
    def load(file_path, datatype=np.float64):
        if file_path.endswith(".bin"):      # legacy: one int32 in front of the (2, D) block
            block = np.fromfile(file_path, dtype=datatype, offset=4).reshape((2, -1))
            mean, std_dev = block[0:1], block[1:2]
        else:
            with np.load(file_path) as archive:
                mean, std_dev = archive["mean"], archive["std_dev"]
        return mean.astype(np.float32, copy=False), std_dev.astype(np.float32, copy=False)
