"""Dict-in / dict-out model wrapper with the reference's interface
(models/NamedForwardModule.py:41-137, models/NamedForwardWrapper.py:19-84): reads
`data[input_names]`, merges them (concatenation on the feature axis by default; sum, mean, product and the
attention-style product summed over time as in NamedForwardModule.merge :115-137, inputs without a time axis or
with a single frame repeated over time first, :139-148), runs the wrapped RNNDyn and writes `data[output_names]`
plus the length dictionaries in place."""
import copy
from functools import reduce
from typing import List

import torch
from torch import nn

from . import rnn_dyn


class NamedForwardWrapper(nn.Module):

    class Config:
        def __init__(self, wrapped_model_config, input_names: List[str], batch_first: bool,
                     input_merge_type: str = "cat", name: str = None,
                     output_names: List[str] = None):
            self.wrapped_model_config = wrapped_model_config
            self.input_names = input_names
            self.batch_first = batch_first
            self.input_merge_type = input_merge_type
            self.name = name
            self.output_names = output_names

        def create_model(self):
            return NamedForwardWrapper(self)

    def __init__(self, config):
        super().__init__()
        self.config = copy.deepcopy(config)
        self.input_names = config.input_names
        self.output_names = config.output_names
        self.batch_first = config.batch_first
        self.name = config.name
        self.input_merge_type = config.input_merge_type
        if config.input_merge_type not in self.MERGE_TYPES:
            raise NotImplementedError("Unknown input merge type {} ({}).".format(
                config.input_merge_type, ", ".join(self.MERGE_TYPES)))
        self.model = config.wrapped_model_config.create_model() \
            if config.wrapped_model_config is not None else None

    MERGE_TYPES = ("cat", "add", "mean", "mul", "attention")      # ModelConfig.MERGE_TYPE_* (:12-17)

    def init_hidden(self, batch_size=1):
        self.model.init_hidden(batch_size)

    @staticmethod
    def _over_time(tensor, length, batch_first):
        """NamedForwardModule._broadcast_time_dim (:139-148): [B, D] gets a time axis, one frame becomes `length`"""
        time_dim = 1 if batch_first else 0
        if tensor.dim() < 3:
            tensor = tensor.unsqueeze(time_dim)
        if tensor.shape[time_dim] == 1 and length != 1:
            tensor = tensor.repeat((1, length, 1) if batch_first else (length, 1, 1))
        return tensor

    @staticmethod
    def merge(inputs, merge_type, batch_first):
        """NamedForwardModule.merge (:115-137) for two or more inputs of equal time extent"""
        if merge_type == "cat":
            return torch.cat(inputs, dim=2)
        if merge_type == "add":
            return torch.stack(inputs).sum(dim=0)
        if merge_type == "mean":
            return torch.stack(inputs).mean(dim=0)
        product = reduce(lambda a, b: a * b, inputs)
        if merge_type == "mul":
            return product
        if merge_type == "attention":       # weights x values, summed over time
            return product.sum(dim=1 if batch_first else 0, keepdim=True)
        raise NotImplementedError("Unknown input merge type {}.".format(merge_type))

    def forward(self, data, lengths, max_lengths, **kwargs):
        inputs = [data[name] for name in self.input_names]
        time_dim = 1 if self.batch_first else 0
        extents = [i.shape[time_dim] for i in inputs if i.dim() > 2]
        longest = max(extents) if extents else 1
        inputs = [self._over_time(i, longest, self.batch_first) for i in inputs]
        input_ = inputs[0] if len(inputs) == 1 else self.merge(inputs, self.input_merge_type, self.batch_first)
        first = self.input_names[0]
        output, kwargs = self.model(input_, seq_lengths_input=lengths[first],
                                    max_length_inputs=max_lengths[first], **kwargs)
        for name in self.output_names:
            data[name] = output
            lengths[name] = kwargs['seq_lengths_input']
            max_lengths[name] = kwargs['max_length_inputs']

    def inference(self, data, lengths, max_lengths, *args, **kwargs):
        self.forward(data, lengths, max_lengths, *args, **kwargs)
