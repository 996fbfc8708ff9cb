"""rnn_dyn.Config / LayerConfig with the reference's fields
(idiaptts/src/neural_networks/pytorch/models/rnn_dyn/Config.py:12-138), restricted to the layer
types on the accelerated path: Linear (+Tanh/ReLU) groups and (Bi)LSTM groups."""
import copy
import re
from typing import List


class Config:

    class LayerConfig:
        layer_type_map = {"FC": "Linear", "linear": "Linear", "LIN": "Linear"}

        def __init__(self, layer_type: str, out_dim: int = None, needs_in_dim: bool = None,
                     num_layers: int = 1, nonlin: str = None, dropout: float = 0.0, **kwargs):
            self.type = self.layer_type_map.get(layer_type, layer_type)
            self.out_dim = out_dim
            self.num_layers = num_layers
            self.dropout = dropout
            self.kwargs = kwargs
            self.nonlin = nonlin
            if nonlin is not None:
                self.nonlin = {"relu": "ReLU", "tanh": "Tanh"}.get(nonlin.lower(), nonlin)
            self.needs_in_dim = needs_in_dim if needs_in_dim is not None else self.type == "Linear"
            self.needs_packing = layer_type in ['LSTM', 'GRU', 'RNN']
            self.needs_transposing = 'Conv' in layer_type or 'BatchNorm' in layer_type

        def __repr__(self):
            rep = "{}x {} {}".format(self.num_layers, self.out_dim if self.out_dim is not None
                                     else "", self.type)
            if self.nonlin is not None:
                rep += " with {}".format(self.nonlin)
            if self.dropout:
                rep += ", dropout: {}".format(self.dropout)
            if self.kwargs:
                rep += ", " + ", ".join("{}: {}".format(k, v) for k, v in self.kwargs.items())
            return rep

    class EmbeddingConfig:
        """An nn.Embedding whose vectors are concatenated to the input of the layer groups named in
        `affected_layer_group_indices` (-1: all groups); the index arrives as a separate input or
        as the last input column (reference Config.py:81-111)."""

        def __init__(self, embedding_dim, name, num_embedding, affected_layer_group_indices=-1):
            self.embedding_dim = embedding_dim
            self.name = name
            self.num_embedding = num_embedding
            if affected_layer_group_indices is None \
                    or isinstance(affected_layer_group_indices, (tuple, list, set)):
                self.affected_layer_group_indices = affected_layer_group_indices
            else:
                self.affected_layer_group_indices = (affected_layer_group_indices,)

        def __repr__(self):
            return "{}: {} inputs, {} embedding dim, affected groups: ({})".format(
                self.name, self.num_embedding, self.embedding_dim,
                ", ".join(str(i) for i in self.affected_layer_group_indices))

    def __init__(self, config_str: str = None, in_dim: int = 0, hparams=None,
                 batch_first: bool = True, layer_configs: List["Config.LayerConfig"] = None,
                 emb_configs=None):
        assert in_dim > 0
        self.in_dim = in_dim
        if config_str is not None:
            self.config_str = config_str
            self.hparams = copy.deepcopy(hparams)
            if self.hparams is not None:
                self.hparams.model_type = config_str
        self.batch_first = batch_first
        self.layer_configs = layer_configs
        self.emb_configs = emb_configs

    def create_model(self):
        from .RNNDyn import RNNDyn
        if self.layer_configs is None:
            cfg = config_from_legacy_string(self.in_dim, self.config_str,
                                            getattr(self.hparams, "batch_first", self.batch_first),
                                            getattr(self.hparams, "dropout", 0.0))
            return RNNDyn(cfg)
        return RNNDyn(self)


def config_from_legacy_string(in_dim, name, batch_first=False, dropout=0.0):
    """'RNNDYN-2_TANH_512-3_BiLSTM_512-1_FC_187' -> Config (reference RNNDyn.py:150-357):
    groups '<n layers>_<type>_<out dim>' separated by '-'; TANH / RELU groups are Linear layers
    each followed by the non-linearity, FC / LIN are linear, a 'Bi' prefix makes an RNN group
    bidirectional; dropout applies inside multi-layer RNN groups and after every FF layer."""
    groups = re.split(r'-\s*(?![^()]*\))', name)[1:]
    if len(groups) == 0:
        raise ValueError("Empty RNNDYN configuration: {}".format(name))
    nonlins = {'RELU': "ReLU", 'TANH': "Tanh"}
    layer_configs = []
    for group in groups:
        attr = group.split('_')
        n_layers, layer_type, out_dim = int(attr[0]), attr[1], int(attr[2])
        bidirectional = layer_type[:2] == 'Bi'
        if bidirectional:
            layer_type = layer_type[2:]
        if layer_type in ('LSTM', 'GRU', 'RNNTANH', 'RNNRELU'):
            rnn_nonlin = {'RNNTANH': 'tanh', 'RNNRELU': 'relu'}.get(layer_type)
            lc = Config.LayerConfig(layer_type='RNN' if rnn_nonlin else layer_type,
                                    out_dim=out_dim, num_layers=n_layers, nonlin=None,
                                    dropout=dropout if n_layers > 1 else 0.0,
                                    bidirectional=bidirectional)
            lc.nonlin = rnn_nonlin                  # reference RNNDyn.py:268-279
            layer_configs.append(lc)
        elif layer_type.upper() in nonlins or layer_type in ('FC', 'LIN', 'linear'):
            layer_configs.append(Config.LayerConfig(layer_type='Linear', out_dim=out_dim,
                                                    num_layers=n_layers,
                                                    nonlin=nonlins.get(layer_type.upper()),
                                                    dropout=dropout))
        else:
            raise NotImplementedError("Layer type {} is outside the accelerated path "
                                      "(SURVEY.md section 2).".format(layer_type))
    return Config(in_dim=int(in_dim), batch_first=batch_first, layer_configs=layer_configs)
