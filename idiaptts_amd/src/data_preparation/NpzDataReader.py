"""Generic reader of the per-stream `.npz` feature archives the label generators write
(reference data_preparation/NpzDataReader.py: DataReader :23-137, NpzDataReader :140-420).

Same constructor surface (`NpzDataReader.Config(name, directory, features, indices,
norm_params_path | norm_params, norm_type, output_names, preprocessing_fn, ...)
.create_reader()`), same order of operations in preprocess_sample / postprocess_sample:

    load         features are looked up by name in `<dir>/<id>.npz` over all directories, cast to
                 float32; one feature -> array, several -> list (in `features` order per directory)
    preprocess   index subset -> [fn] -> normalise -> [fn] -> float32
    postprocess  [fn] -> de-normalise -> [fn]

The host logic is index / file work; the heavy consumer of these
readers, the training input pipeline, can bypass the per-item path entirely through FrameShard
(`hparams.resident_dataset`), which calls `load` + `preprocess_sample` once per utterance.
"""
import os
from enum import Enum

import numpy as np

from ...misc.normalisation.MeanCovarianceExtractor import MeanCovarianceExtractor
from ...misc.normalisation.MeanStdDevExtractor import MeanStdDevExtractor
from ...misc.normalisation.MinMaxExtractor import MinMaxExtractor
from .DataReaders import ReaderBase


class DataReader(ReaderBase):
    """Base of all readers: `reader[id]` -> {output_name: sample, "_id_list": id}."""

    class Config(object):
        def __init__(self, name, chunk_size=1, match_length=None, output_names=None,
                     random_select=True, max_frames=None, min_frames=None, pad_mode='constant',
                     other_pad_dims=None, requires_seq_mask=False):
            self.name = name
            self.chunk_size = chunk_size
            if match_length is not None and not isinstance(match_length, (tuple, list)):
                match_length = (match_length,)
            self.match_length = match_length
            if output_names is None:
                output_names = (name,)
            elif not isinstance(output_names, (tuple, list)):
                output_names = (output_names,)
            self.output_names = output_names
            self.random_select = random_select
            self.max_frames, self.min_frames = max_frames, min_frames
            self.pad_mode, self.other_pad_dims = pad_mode, other_pad_dims
            self.requires_seq_mask = requires_seq_mask

    def __init__(self, config):
        self._configure(config.name, output_names=config.output_names,
                        match_length=config.match_length, min_frames=config.min_frames,
                        max_frames=config.max_frames, pad_mode=config.pad_mode,
                        other_pad_dims=config.other_pad_dims, random_select=config.random_select,
                        chunk_size=config.chunk_size, requires_seq_mask=config.requires_seq_mask)

    def load(self, id_name):
        raise NotImplementedError("Class {} doesn't implement load(id_name)."
                                  .format(self.__class__.__name__))

    def preprocess_sample(self, sample):
        raise NotImplementedError("Class {} doesn't implement preprocess_sample(sample)."
                                  .format(self.__class__.__name__))


class NpzDataReader(DataReader):

    class Config(DataReader.Config):

        class NormType(Enum):
            NONE = "None"
            MEAN_VARIANCE = "mean_variance"
            MEAN_STDDEV = "mean_stddev"
            MIN_MAX = "min_max"

        def __init__(self, name, directory=None, features=None, indices=None,
                     norm_params_path=None, norm_params=None, norm_type=NormType.NONE,
                     output_names=None, preprocessing_fn=None, preprocess_before_norm=False,
                     postprocessing_fn=None, postprocess_before_norm=True, **kwargs):
            features = name if features is None else features
            self.features = features if isinstance(features, list) else [features]
            if indices is not None and not isinstance(indices, dict):
                indices = np.asarray(indices).astype(np.int64)
            self.indices = indices
            super().__init__(name=name,
                             output_names=self.features if output_names is None else output_names,
                             **kwargs)
            self.directory = list(directory) if isinstance(directory, (tuple, list)) \
                else [directory]
            self.norm_params_path = norm_params_path
            self.norm_params = norm_params
            self.norm_type = norm_type
            self.preprocessing_fn = preprocessing_fn
            self.preprocess_before_norm = preprocess_before_norm
            self.postprocessing_fn = postprocessing_fn
            self.postprocess_before_norm = postprocess_before_norm

        def create_reader(self):
            return NpzDataReader(self)

    _NORMALISERS = {Config.NormType.MEAN_VARIANCE: MeanCovarianceExtractor,
                    Config.NormType.MEAN_STDDEV: MeanStdDevExtractor,
                    Config.NormType.MIN_MAX: MinMaxExtractor}

    def __init__(self, config):
        super().__init__(config)
        self.directory = config.directory
        self.features = config.features
        self.indices = config.indices
        if config.norm_type == NpzDataReader.Config.NormType.NONE:
            self.normaliser = None
        elif config.norm_type in self._NORMALISERS:
            self.normaliser = self._NORMALISERS[config.norm_type]()
        else:
            raise NotImplementedError("Unknown norm_type {}".format(config.norm_type))
        if config.norm_params is not None:
            self.norm_params = config.norm_params
        elif config.norm_params_path is not None:
            self.norm_params = self.normaliser.load(config.norm_params_path)
        else:
            self.norm_params = None
        self.preprocessing_fn = config.preprocessing_fn
        self.preprocess_before_norm = config.preprocess_before_norm
        self.postprocessing_fn = config.postprocessing_fn
        self.postprocess_before_norm = config.postprocess_before_norm

    # ---------------------------------------------------------------- normalisation parameters
    def get_normalisation_params(self, dir_out=None, file_name=None):
        """Loads `<file_name->mean-std_dev | mean-covariance | min-max` (.npz, legacy .bin) from
        `dir_out`, or from every feature directory that has one (several hits -> a list, one
        per directory)."""
        if self.normaliser is None:
            return self.norm_params
        if dir_out is not None:
            self.norm_params = self._load_normalisation_params(dir_out, file_name)
            return self.norm_params
        found = []
        for directory in self.directory:
            try:
                found.append(self._load_normalisation_params(directory, file_name))
            except FileNotFoundError:
                pass
        assert len(found) > 0, "No normalisation parameter file found in self.directory and " \
                               "dir_out was None."
        self.norm_params = found[0] if len(found) == 1 else found
        return self.norm_params

    def _load_normalisation_params(self, directory, file_name=None):
        prefix = ""
        if file_name is not None:
            prefix = file_name + ("-" if os.path.basename(file_name) != "" else "")
        base = os.path.join(directory, prefix + self.normaliser.file_name_appendix)
        try:
            return self.normaliser.load(base + ".npz")
        except FileNotFoundError:
            return self.normaliser.load(base + ".bin")

    def _params_of(self, feature_idx):
        if self.norm_params is None:
            raise ValueError("norm_params not set, call get_normalisation_params() before.")
        if isinstance(self.norm_params[0], (tuple, list)):
            return self.norm_params[feature_idx]
        return self.norm_params

    # ------------------------------------------------------------------------------- samples
    def load(self, id_name):
        id_name = os.path.splitext(os.path.basename(id_name))[0]
        missing = list(self.features)
        loaded = []
        for directory in self.directory:
            path = os.path.join(directory, id_name + ".npz")
            if not os.path.isfile(path):
                continue
            with np.load(path) as archive:
                here = [n for n in missing if n in archive]
                loaded += [archive[n].astype(np.float32, copy=False) for n in here]
            missing = [n for n in missing if n not in here]
        if missing:
            raise FileNotFoundError("Cannot find file {}.npz or features {} in it in [{}]".format(
                id_name, ", ".join(missing), ",".join(str(d) for d in self.directory)))
        return loaded[0] if len(loaded) == 1 else loaded

    def preprocess_sample(self, features, feature_idx=0):
        if isinstance(features, list):
            return [self.preprocess_sample(f, feature_idx=i) for i, f in enumerate(features)]
        if self.indices is not None:
            features = self._get_features_subset(features)
        if self.preprocess_before_norm and self.preprocessing_fn is not None:
            features = self.preprocessing_fn(features)
        if self.normaliser is not None:
            assert not isinstance(features, (tuple, list)), "Multiple features not supported."
            features = self.normaliser._normalise(features, *self._params_of(feature_idx))
        if not self.preprocess_before_norm and self.preprocessing_fn is not None:
            features = self.preprocessing_fn(features)
        return features.astype(np.float32, copy=False)

    def _get_features_subset(self, features):
        if isinstance(self.indices, dict):
            return features[tuple(self.indices.get(dim, slice(None))
                                  for dim in range(features.ndim))]
        return features[..., self.indices]

    def postprocess_sample(self, features, feature_idx=0):
        if isinstance(features, dict):
            return {name: self.postprocess_sample(features[name], feature_idx=i)
                    for i, name in enumerate(self.features)}
        if self.postprocess_before_norm and self.postprocessing_fn is not None:
            features = self.postprocessing_fn(features)
        if self.norm_params is not None:
            assert not isinstance(features, (tuple, list)), "Multiple features not supported."
            features = self.normaliser._denormalise(features, *self._params_of(feature_idx))
        if not self.postprocess_before_norm and self.postprocessing_fn is not None:
            features = self.postprocessing_fn(features)
        return features
