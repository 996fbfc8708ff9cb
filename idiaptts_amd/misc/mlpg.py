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

    def generation_streams(self, matrices, streams, device=None):
        """Several MLPG problems on the SAME rows in one host -> device -> host round trip: `matrices` is a list of
        [T_u, C] arrays (the de-normalised network outputs of a batch of utterances), `streams` a list of
        (first column, covariance [3 D, 3 D], D).  The rows go to the device once (in their own dtype; float32 stays float32, anything else becomes float64 there),
        every stream is one launch over all utterances reading its columns in place, all trajectories come back in
        one [sum T_u, sum D] copy.  Returns, per stream, the list of [T_u, D] float64 trajectories (views of that
        copy) -- what `generation_batch` returns for the stream's column block."""
        _lib.require_gpu()
        dev = torch.device(device if device is not None else "cuda")
        lengths = [m.shape[0] for m in matrices]
        off = [0]
        for n in lengths:
            off.append(off[-1] + n)
        host = np.concatenate([np.asarray(m) for m in matrices], axis=0) if len(matrices) > 1 else np.asarray(matrices[0])
        feats = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
        if feats.dtype != torch.float32:          # (float32 rows are widened in the solve's loads)
            feats = feats.double()
        total = sum(int(d) for _, _, d in streams)
        out = torch.empty((off[-1], total), dtype=torch.float64, device=dev)
        o0 = 0
        plan = ops.MlpgPlan(off) if len(streams) > 1 else None      # the offsets' share of a call, once for all streams
        for col0, covariance, dim in streams:
            var = np.ascontiguousarray(np.diag(np.asarray(covariance))[:3 * dim], dtype=np.float64)
            ops.mlpg_generation(feats, torch.from_numpy(var).to(dev), int(dim), off, col0=int(col0), out=out, ocol0=o0,
                                plan=plan)
            o0 += int(dim)
        res = out.cpu().numpy()             # (synchronises: the plan's table is no longer read)
        if plan is not None:
            plan.close()
        result, o0 = [], 0
        for _, _, dim in streams:
            result.append([res[off[u]:off[u + 1], o0:o0 + dim] for u in range(len(lengths))])
            o0 += int(dim)
        return result
