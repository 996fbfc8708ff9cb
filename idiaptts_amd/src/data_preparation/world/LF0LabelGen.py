"""LF0 + V/UV labels from audio (reference data_preparation/world/LF0LabelGen.py:32-322):
`pyworld.dio` + `pyworld.stonemask` (:263-264) on the HIP DIO / StoneMask kernels, all utterances
of a batch in one launch sequence; log-F0 with the 20 Hz silence threshold, linear interpolation
through unvoiced stretches, legacy raw-float32 files (`lf0/<id>.lf0`, `vuv/<id>.vuv`, with deltas
`lf0/<id>.lf0_deltas` = [lf0, d, dd, vuv]) and mean / std-dev normalisation (V/UV fixed at 0 / 1).
SURVEY.md section 8(f) row 4."""
import glob
import logging
import math
import os
from collections import OrderedDict

import numpy as np

from idiaptts_amd import world as _world
from idiaptts_amd.misc.normalisation.MeanStdDevExtractor import MeanStdDevExtractor
from idiaptts_amd.misc.utils import compute_deltas, interpolate_lin
from idiaptts_amd.src.data_preparation.audio.AudioProcessing import AudioProcessing


class LF0LabelGen(object):
    f0_silence_threshold = 20
    # "dio" (dio + stonemask, the reference: :263-264) or "harvest"; an extension, see WorldFeatLabelGen
    f0_estimator = "dio"
    lf0_zero = 0

    dir_lf0 = "lf0"
    dir_deltas = "lf0"
    dir_vuv = "vuv"
    ext_lf0 = ".lf0"
    ext_deltas = ".lf0_deltas"
    ext_vuv = ".vuv"

    logger = logging.getLogger(__name__)

    def __init__(self, dir_labels, add_deltas=False, batch_utts=32):
        self.dir_labels = dir_labels
        self.add_deltas = add_deltas
        self.norm_params = None
        self.batch_utts = batch_utts

    def __getitem__(self, id_name):
        return self.preprocess_sample(self.load_sample(id_name, self.dir_labels, self.add_deltas))

    @staticmethod
    def trim_end_sample(sample, length, reverse=False):
        if length == 0:
            return sample
        return sample[length:, ...] if reverse else sample[:-length, ...]

    def _params(self, norm_params):
        params = norm_params if norm_params is not None else self.norm_params
        if params is None:
            raise ValueError("Give norm_params or call get_normalisation_params() before.")
        return params

    def preprocess_sample(self, sample, norm_params=None):
        mean, std_dev = self._params(norm_params)
        return np.float32((sample - mean) / std_dev)

    def postprocess_sample(self, sample, norm_params=None):
        mean, std_dev = self._params(norm_params)
        return np.copy((sample * std_dev) + mean)

    @staticmethod
    def load_lf0(id_name, dir_out, add_deltas=False):
        ext, width = (LF0LabelGen.ext_deltas, 3) if add_deltas else (LF0LabelGen.ext_lf0, 1)
        return np.fromfile(os.path.join(dir_out, LF0LabelGen.dir_lf0, id_name + ext),
                           dtype=np.float32).reshape(-1, width)

    @staticmethod
    def load_vuv(id_name, dir_out):
        return np.fromfile(os.path.join(dir_out, LF0LabelGen.dir_vuv, id_name + LF0LabelGen.ext_vuv),
                           dtype=np.float32).reshape(-1, 1)

    @staticmethod
    def load_sample(id_name, dir_out, add_deltas=False):
        """[frames, (lf0 | lf0, d, dd), vuv]"""
        if add_deltas:           # gen_data stores the four columns in one file
            return np.fromfile(os.path.join(dir_out, LF0LabelGen.dir_deltas,
                                            id_name + LF0LabelGen.ext_deltas),
                               dtype=np.float32).reshape(-1, 4)
        return np.concatenate((LF0LabelGen.load_lf0(id_name, dir_out),
                               LF0LabelGen.load_vuv(id_name, dir_out)), axis=1)

    @staticmethod
    def convert_to_world_features(sample):
        lf0 = sample[:, 0]
        vuv = np.copy(sample[:, -1])
        vuv[vuv < 0.5] = 0.0
        vuv[vuv >= 0.5] = 1.0
        return lf0, vuv

    def get_normalisation_params(self, dir_out, file_name=None):
        name = (file_name + "-" if file_name is not None else "") \
            + MeanStdDevExtractor.file_name_appendix
        path = os.path.join(dir_out, self.dir_deltas if self.add_deltas else self.dir_lf0, name)
        path += ".npz" if os.path.isfile(path + ".npz") else ".bin"
        mean, std_dev = MeanStdDevExtractor.load(path)
        if self.add_deltas:
            self.norm_params = (mean, std_dev)
        else:
            self.norm_params = (np.concatenate((np.atleast_2d(mean), np.atleast_2d(0.0)), axis=1),
                                np.concatenate((np.atleast_2d(std_dev), np.atleast_2d(1.0)), axis=1))
        return self.norm_params

    @staticmethod
    def extract_batch(raws, fs, hop_size_ms=5):
        """[(lf0 [T,1] float32 interpolated, vuv [T,1]) ...] for waveforms of one sampling rate."""
        out = []
        for feats in _world.analyse_batch(raws, fs, hop_size_ms, want_sp=False, want_bap=False,
                                          f0_method=LF0LabelGen.f0_estimator):
            f0 = feats["f0"]
            with np.errstate(divide="ignore"):
                lf0 = np.log(f0, dtype=np.float32) if f0.dtype == np.float32 \
                    else np.log(f0).astype(np.float32)
            lf0[lf0 <= math.log(LF0LabelGen.f0_silence_threshold)] = LF0LabelGen.lf0_zero
            out.append(interpolate_lin(lf0))
        return out

    def gen_data(self, dir_in, dir_out=None, file_id_list="", id_list=None, add_deltas=False,
                 return_dict=False):
        if id_list is None:
            id_list = [os.path.splitext(os.path.basename(f))[0]
                       for f in glob.glob(os.path.join(dir_in, "*.wav"))]
            file_id_list_name = "all"
        else:
            file_id_list_name = os.path.splitext(os.path.basename(file_id_list))[0]
        if dir_out is not None:
            for d in ([self.dir_deltas] if add_deltas else [self.dir_lf0, self.dir_vuv]):
                os.makedirs(os.path.join(dir_out, d), exist_ok=True)
        label_dict = OrderedDict()
        extractor = MeanStdDevExtractor()
        for b0 in range(0, len(id_list), self.batch_utts):
            names = id_list[b0:b0 + self.batch_utts]
            raws, fss = zip(*[AudioProcessing.get_raw(os.path.join(dir_in, n + ".wav"))
                              for n in names])
            assert len(set(fss)) == 1, "All files of a batch need the same sampling rate."
            for name, (lf0, vuv) in zip(names, self.extract_batch(list(raws), fss[0])):
                vuv = vuv.astype(np.float32)
                if add_deltas:
                    deltas = compute_deltas(lf0)
                    labels = np.concatenate((lf0, deltas, compute_deltas(deltas), vuv), axis=1)
                    if dir_out is not None:
                        labels.tofile(os.path.join(dir_out, self.dir_deltas, name + self.ext_deltas))
                    extractor.add_sample(labels)
                else:
                    labels = np.concatenate((lf0, vuv), axis=1)
                    if dir_out is not None:
                        lf0.tofile(os.path.join(dir_out, self.dir_lf0, name + self.ext_lf0))
                        vuv.tofile(os.path.join(dir_out, self.dir_vuv, name + self.ext_vuv))
                    extractor.add_sample(lf0)
                if return_dict:
                    label_dict[name] = labels
        if add_deltas:           # V/UV column: mean 0, variance 1 by definition (reference :300-303)
            extractor.sum_frames[..., -1] = 0.0
            extractor.sum_squared_frames[..., -1] = extractor.sum_length
        if dir_out is not None:
            extractor.save(os.path.join(dir_out, self.dir_deltas if add_deltas else self.dir_lf0,
                                        file_id_list_name))
        mean, std_dev = extractor.get_params()
        if not add_deltas:
            mean = np.concatenate((np.atleast_1d(np.squeeze(mean)), (0.0,)), axis=0)
            std_dev = np.concatenate((np.atleast_1d(np.squeeze(std_dev)), (1.0,)), axis=0)
        if return_dict:
            return label_dict, mean, std_dev
        return mean, std_dev
