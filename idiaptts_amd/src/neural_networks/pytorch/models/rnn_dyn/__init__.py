from .Config import Config, config_from_legacy_string  # noqa: F401
from .RNNDyn import FFWrapper, RNNDyn, RNNWrapper  # noqa: F401


def convert_legacy_to_config(in_dim, hparams):
    """reference rnn_dyn/__init__: legacy model_type string -> Config."""
    dim = in_dim[0] if isinstance(in_dim, (tuple, list)) else in_dim
    return config_from_legacy_string(dim, hparams.model_type,
                                     getattr(hparams, "batch_first", False),
                                     getattr(hparams, "dropout", 0.0))
