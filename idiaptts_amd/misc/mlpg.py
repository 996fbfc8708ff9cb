"""MLPG with the reference's interface (idiaptts/misc/mlpg.py:28-127), computed by the HIP
pentadiagonal solver (idiaptts_amd/csrc/mlpg.hip) instead of 62 bandmat calls per utterance."""
import numpy as np
import torch

from .. import lib as _lib
from .. import ops


class MLPG(object):

    def generation(self, features, covariance, feature_dim):
        """features [T, 3*feature_dim] (static | delta | delta-delta), covariance
        [3*feature_dim, 3*feature_dim] -> smoothed trajectory [T, feature_dim] float64."""
        return self.generation_batch([features], covariance, feature_dim)[0]

    def generation_batch(self, features_list, covariance, feature_dim, device=None):
        _lib.require_gpu()
        dev = torch.device(device if device is not None else "cuda")
        covariance = np.asarray(covariance)
        var = np.ascontiguousarray(np.diag(covariance)[:3 * feature_dim], dtype=np.float64)
        lengths = [f.shape[0] for f in features_list]
        off = [0]
        for n in lengths:
            off.append(off[-1] + n)
        feats = np.ascontiguousarray(
            np.concatenate([np.asarray(f)[:, :3 * feature_dim] for f in features_list], axis=0),
            dtype=np.float64)
        out = ops.mlpg_generation(torch.from_numpy(feats).to(dev), torch.from_numpy(var).to(dev),
                                  feature_dim, off).cpu().numpy()
        return [out[off[u]:off[u + 1]] for u in range(len(lengths))]
