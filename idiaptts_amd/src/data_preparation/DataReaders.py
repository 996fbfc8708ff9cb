"""Reader-side helpers of the hot path: per-stream `.npz` / legacy raw-float32 loading with
(de)normalisation (reference NpzDataReader.py:312-420, QuestionLabelGen reader) and the length
matching of PyTorchDatareadersDataset (:99-197) -- symmetric trimming, bit-exact index math."""
import os

import numpy as np


def trim_to_reference(value, ref_lengths):
    """PyTorchDatareadersDataset._trim_datareader_output (:179-197): for every leading dim,
    front = (-diff) // 2, end = -diff - front; raises ValueError when the reference is longer.
    Returns (trimmed value, was_trimmed)."""
    slices, do_trim = [], False
    for dim, ref_length in enumerate(ref_lengths):
        len_diff = ref_length - value.shape[dim]
        if len_diff > 0:
            raise ValueError()
        front = (-len_diff) // 2
        end = -len_diff - front
        slices.append(slice(front, value.shape[dim] - end))
        do_trim |= front != 0 or end != 0
    return (value[tuple(slices)], True) if do_trim else (value, False)


def match_lengths(outputs, match_length):
    """outputs: {name: array}; match_length: {name: [reference names]} as in the reader configs.
    Repeats trimming until stable (cyclic dependencies allowed), like _match_output_lengths
    (:99-140): a stream is trimmed to its reference; if the reference is longer the pair is
    skipped and the reference gets trimmed on its own turn."""
    changed = True
    while changed:
        changed = False
        for name, refs in match_length.items():
            refs = [r for r in refs if r in outputs] if refs else refs   # streams not loaded now
            if not refs:
                continue
            ref_lengths = [outputs[r].shape[0] for r in refs]
            try:
                outputs[name], was = trim_to_reference(outputs[name], ref_lengths)
            except ValueError:
                was = False
            if was:
                changed = True
                break
    return outputs


class NpzStreamReader(object):
    """Loads `<dir>/<id>.npz[feature_names]` (concatenated on the feature axis) or a legacy raw
    float32 file `<dir>/<id>.<ext>` with `legacy_dim` columns; optional mean/std or min/max
    normalisation as in NpzDataReader.preprocess_sample / postprocess_sample (:347-420)."""

    def __init__(self, directory, feature_names, legacy_ext=None, legacy_dim=None,
                 norm_type=None, norm_params=None):
        self.directory = directory
        self.feature_names = [feature_names] if isinstance(feature_names, str) else feature_names
        self.legacy_ext, self.legacy_dim = legacy_ext, legacy_dim
        self.norm_type, self.norm_params = norm_type, norm_params

    def load(self, id_name):
        base = os.path.join(self.directory, os.path.basename(id_name))
        if os.path.isfile(base + ".npz"):
            arch = np.load(base + ".npz")
            feats = [arch[n] for n in self.feature_names]
            return feats[0] if len(feats) == 1 else np.concatenate(feats, axis=1)
        if self.legacy_ext is None:
            raise FileNotFoundError(base + ".npz")
        return np.fromfile(base + "." + self.legacy_ext, dtype=np.float32) \
            .reshape(-1, self.legacy_dim)

    def normalise(self, sample):
        if self.norm_type is None:
            return sample
        a, b = self.norm_params
        if self.norm_type == "mean_variance":
            return ((sample - a) / b).astype(np.float32)
        if self.norm_type == "min_max":     # MinMaxExtractor._normalise (:35-38): (x-min)/range,
            rng = np.array(b - a)           # zero ranges replaced by 1 (_fix_range_inplace)
            rng[rng == 0.0] = 1.0
            return ((sample - a) / rng).astype(np.float32)
        raise NotImplementedError(self.norm_type)

    def denormalise(self, sample):
        if self.norm_type is None:
            return sample
        a, b = self.norm_params
        if self.norm_type == "mean_variance":
            return sample * b + a
        if self.norm_type == "min_max":
            rng = np.array(b - a)
            rng[rng == 0.0] = 1.0
            return sample * rng + a
        raise NotImplementedError(self.norm_type)


class ReaderBase(object):
    """What every data reader exposes to the dataset and to prepare_batch (reference
    data_preparation/DataReader.py:22-137 plus the attributes DataReaderConfig.create_reader
    attaches, DataReaderConfig.py:120-143): `reader[id]` -> {output_name: preprocessed sample,
    "_id_list": id}, `get_length`, `pad`, `trim`."""

    name = None
    output_names = None
    match_length = None
    min_frames = None
    max_frames = None
    pad_mode = 'constant'
    other_pad_dims = None
    random_select = False
    chunk_size = 1
    requires_seq_mask = False

    def _configure(self, name, output_names=None, match_length=None, min_frames=None,
                   max_frames=None, pad_mode='constant', other_pad_dims=None,
                   random_select=False, chunk_size=1, requires_seq_mask=False):
        self.name = name
        self.output_names = list(output_names) if output_names is not None else [name]
        if match_length is not None and not isinstance(match_length, (tuple, list)):
            match_length = (match_length,)
        self.match_length = match_length
        self.min_frames, self.max_frames = min_frames, max_frames
        self.pad_mode, self.other_pad_dims = pad_mode, other_pad_dims
        self.random_select, self.chunk_size = random_select, chunk_size
        self.requires_seq_mask = requires_seq_mask
        self._length_cache = {}
        return self

    def __getitem__(self, id_name):
        item = self.preprocess_sample(self.load(id_name))
        items = item if isinstance(item, (tuple, list)) else (item,)
        if len(items) != len(self.output_names):
            raise RuntimeError("The data reader returns {} item(s) but {} output names were given."
                               .format(len(items), len(self.output_names)))
        if self.chunk_size > 1:
            items = [self.pad(i, chunk_padding(i, self.chunk_size)) for i in items]
        out = dict(zip(self.output_names, items))
        out["_id_list"] = id_name
        if id_name not in self._length_cache:
            # what get_length would work out from this very item: the dataset asks for the length of every reader it
            # matches another one to right after loading both, and used to load each of them a second time for it
            c = self.chunk_size
            try:
                self._length_cache[id_name] = ((max(len(v) for v in items) + c - 1) // c) * c
            except TypeError:        # an output without a length: get_length will say so if it is ever asked
                pass
        return out

    def get_length(self, id_name):
        if id_name not in self._length_cache:
            out = self[id_name]
            length = max(len(v) for k, v in out.items() if k != "_id_list")
            c = self.chunk_size
            self._length_cache[id_name] = ((length + c - 1) // c) * c
        return self._length_cache[id_name]

    def pad(self, sample, pad_width, pad_mode=None):
        return np.pad(sample, pad_width, self.pad_mode if pad_mode is None else pad_mode)

    @staticmethod
    def trim(sample, trim_width):
        """trim_width: per dim (front, end) tuples (DataReader.trim :131-137)."""
        if (np.array(trim_width) == 0).all():
            return sample
        return sample[tuple(slice(v[0], sample.shape[d] - v[1]) if isinstance(v, tuple) else v
                            for d, v in enumerate(trim_width))]

    @staticmethod
    def trim_end_sample(sample, length, reverse=False):
        """Removes `length` frames from the end (reverse: from the front), as the reference's
        docstring states (DataReader.py:118-129).  The reference's body hands plain ints to
        `trim`, which then indexes a single element instead of slicing -- its only caller
        (AcousticModelTrainer.synthesize :497-500, the synth_load_org_* path) cannot work as
        written, so the documented behaviour is what is implemented here."""
        if length == 0:
            return sample
        return sample[length:] if reverse else sample[:len(sample) - length]


def chunk_padding(item, chunk_size):
    length = len(item)
    first = (0, ((length + chunk_size - 1) // chunk_size) * chunk_size - length)
    return (first, *([(0, 0)] * (item.ndim - 1)))
