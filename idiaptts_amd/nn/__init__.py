from .modules import GRU, LSTM, RNN, LinearAct  # noqa: F401
