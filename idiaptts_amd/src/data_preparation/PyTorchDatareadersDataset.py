"""Map-style dataset over a set of data readers (reference
data_preparation/PyTorchDatareadersDataset.py:20-197): item = ({output_name: sample, ...,
"_id_list": id}, dataset) with every reader's outputs trimmed symmetrically to the readers named
in its `match_length` (front = diff // 2, end = diff - front; integer index math, bit-exact)."""
from typing import List

import torch
from torch.utils.data import Dataset

from idiaptts_amd.src.data_preparation.DataReaders import trim_to_reference


class PyTorchDatareadersDataset(Dataset):

    def __init__(self, id_list: List[str], datareaders: List, *args, **kwargs):
        self.id_list = id_list
        # The reference keeps a set; a list keeps the key order of the batch dicts deterministic.
        self.datareaders = list(dict.fromkeys(datareaders))

    def get_input_dim(self, input_names=None):
        output_dict, _ = self[0]
        return sum(output_dict[name].shape[1] for name in input_names)

    def __len__(self):
        return len(self.id_list)

    def __getitem__(self, item):
        return self.get_id_name(self.id_list[item])

    def get_id_name(self, id_name):
        output_dict = {}
        for reader in self.datareaders:
            reader_output = reader[id_name]
            for key in reader_output:
                if key != "_id_list" and key in output_dict:
                    raise KeyError("Feature {} defined twice.".format(key))
            output_dict.update(reader_output)
        self._match_output_lengths(output_dict, id_name)
        self._match_max_frames(output_dict, id_name)
        return output_dict, self

    # ------------------------------------------------------------------ length matching
    def _match_output_lengths(self, output_dict, id_name):
        known_length = {}
        was_trimmed = True
        while was_trimmed:
            was_trimmed = False
            for reader in self.datareaders:
                if reader.match_length is not None and \
                        self._trim_datareader(reader, output_dict, id_name, known_length):
                    was_trimmed = True
                    break

    def _trim_datareader(self, reader, output_dict, id_name, known_length):
        ref_lengths = self._get_ref_lengths(reader.match_length, id_name, known_length)
        if not ref_lengths:
            return False
        for key in reader.output_names:
            if key == "_id_list":
                continue
            try:
                output_dict[key], was_trimmed = trim_to_reference(output_dict[key], ref_lengths)
            except ValueError:       # reference feature is longer: it is trimmed on its turn
                continue
            if was_trimmed:
                known_length[reader.name] = ref_lengths
                return True
        return False

    def _get_ref_lengths(self, match_length, id_name, known_length):
        ref_lengths = []
        present = {r.name for r in self.datareaders}
        for name in match_length:
            if name not in present:      # stream not loaded (inference without stored targets)
                continue
            reader = self.get_datareader_by_name(name)
            if reader.name not in known_length:
                known_length[reader.name] = [reader.get_length(id_name)]
            ref_lengths.append(known_length[reader.name][0])
        return ref_lengths

    def _match_max_frames(self, output_dict, id_name):
        """A reader with `max_frames` keeps one window of that many frames (random start with
        `random_select`); the same window is applied along its `match_length` chain to readers
        that also set `max_frames`, cycles included (reference :199-257)."""
        processed = set()
        for reader in self.datareaders:
            if reader.max_frames is None:
                continue
            feature_len = reader.get_length(id_name)
            if feature_len <= reader.max_frames:
                return
            start = 0
            if reader.random_select:
                start = int(torch.randint(0, max(1, feature_len - reader.max_frames), (1,)))
            self._select_max_frames(processed, output_dict, reader, start,
                                    min(start + reader.max_frames, feature_len))

    def _select_max_frames(self, processed, output_dict, reader, start, end):
        processed.add(reader)
        for name in reader.output_names:
            output_dict[name] = output_dict[name][start:end]
        present = {r.name for r in self.datareaders}
        for ref_name in reader.match_length or ():
            if ref_name not in present:
                continue
            ref = self.get_datareader_by_name(ref_name)
            if ref.max_frames is not None and ref not in processed:
                self._select_max_frames(processed, output_dict, ref, start, end)

    def get_datareader_by_name(self, name):
        for reader in self.datareaders:
            if name == reader.name:
                return reader
        raise KeyError("No data reader named {} found in {}.".format(
            name, [r.name for r in self.datareaders]))

    def get_datareader_by_output_name(self, name):
        for reader in self.datareaders:
            if name in reader.output_names:
                return reader
        raise KeyError("No data reader with output name {} found in {}.".format(
            name, [r.name for r in self.datareaders]))
