"""Dict-in / dict-out model wrapper with the reference's interface
(models/NamedForwardModule.py:41-137, models/NamedForwardWrapper.py:19-84): reads
`data[input_names]`, concatenates them on the feature axis, runs the wrapped RNNDyn and writes
`data[output_names]` plus the length dictionaries in place."""
import copy
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
        if config.input_merge_type != "cat":
            raise NotImplementedError("Only MERGE_TYPE_CAT is on the accelerated path.")
        self.model = config.wrapped_model_config.create_model() \
            if config.wrapped_model_config is not None else None

    def init_hidden(self, batch_size=1):
        self.model.init_hidden(batch_size)

    def forward(self, data, lengths, max_lengths, **kwargs):
        inputs = [data[name] for name in self.input_names]
        input_ = inputs[0] if len(inputs) == 1 else torch.cat(inputs, dim=2)
        first = self.input_names[0]
        output, kwargs = self.model(input_, seq_lengths_input=lengths[first],
                                    max_length_inputs=max_lengths[first], **kwargs)
        for name in self.output_names:
            data[name] = output
            lengths[name] = kwargs['seq_lengths_input']
            max_lengths[name] = kwargs['max_length_inputs']

    def inference(self, data, lengths, max_lengths, *args, **kwargs):
        self.forward(data, lengths, max_lengths, *args, **kwargs)
