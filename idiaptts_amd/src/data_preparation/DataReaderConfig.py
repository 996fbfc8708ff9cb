"""DataReaderConfig: declarative description of one data reader, as the trainers receive them
(reference data_preparation/DataReaderConfig.py:34-143).  `create_reader` instantiates the
reader class named by `feature_type`, loads its normalisation parameters and attaches the
name / output names / length-matching attributes the dataset and prepare_batch rely on."""
import os
from typing import List, Union


class DataReaderConfig(object):

    def __init__(self, name: str, feature_type, directory: Union[str, os.PathLike] = None,
                 features: Union[str, List[str]] = None, output_names: List[str] = None,
                 match_length: Union[str, List[str]] = None, min_frames: int = None,
                 max_frames: int = None, pad_mode: str = 'constant',
                 other_pad_dims: List[int] = None, random_select: bool = False,
                 chunk_size: int = 1, requires_seq_mask: bool = False, **kwargs):
        self.name = name
        self.type = feature_type
        self.directory = directory
        self.features = self._str_to_list(name if features is None else features)
        self.output_names = self.features if output_names is None else output_names
        self.match_length = match_length if type(match_length) in (tuple, list) \
            or match_length is None else (match_length,)
        self.min_frames, self.max_frames = min_frames, max_frames
        self.pad_mode, self.other_pad_dims = pad_mode, other_pad_dims
        self.random_select, self.chunk_size = random_select, chunk_size
        self.requires_seq_mask = requires_seq_mask
        self.kwargs = kwargs

    @staticmethod
    def _str_to_list(string):
        return string if type(string) is list else [string]

    def _reader_class(self):
        if not isinstance(self.type, str):
            return self.type
        if self.type == "QuestionLabelGen":
            from idiaptts_amd.src.data_preparation.questions.QuestionLabelGen import \
                QuestionLabelGen
            return QuestionLabelGen
        if self.type == "WorldFeatLabelGen":
            from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import \
                WorldFeatLabelGen
            return WorldFeatLabelGen
        if self.type == "PhonemeDurationLabelGen":
            from idiaptts_amd.src.data_preparation.phonemes.PhonemeDurationLabelGen import \
                PhonemeDurationLabelGen
            return PhonemeDurationLabelGen
        raise NotImplementedError("Unknown data reader type {}.".format(self.type))

    def create_reader(self):
        reader = self._reader_class()(dir_labels=self.directory, **self.kwargs)
        if callable(getattr(reader, "get_normalisation_params", None)):
            reader.get_normalisation_params()
        reader._configure(self.name, self.output_names, self.match_length, self.min_frames,
                          self.max_frames, self.pad_mode, self.other_pad_dims,
                          self.random_select, self.chunk_size, self.requires_seq_mask)
        return reader
