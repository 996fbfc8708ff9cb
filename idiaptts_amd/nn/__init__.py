from .modules import GRU, LSTM, LinearAct  # noqa: F401
