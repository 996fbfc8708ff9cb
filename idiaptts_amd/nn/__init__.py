from .modules import LSTM, LinearAct  # noqa: F401
