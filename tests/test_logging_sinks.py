"""idiaptts_amd/misc/logging_sinks.py: the scalar writer the handler and trainer log through (reference
ModularModelHandlerPyTorch.py:694-705, 858-867; ModularTrainer.py:198-214), the memory probe that replaces
`nvidia-smi` (misc/utils.py:152-175) and the NaN / Inf guard that is looked at one step late (:778-781)."""
import json
import os

import pytest
import torch

from idiaptts_amd.misc import logging_sinks
from idiaptts_amd.src.ExtendedHParams import ExtendedHParams


def _hparams(tmp_path):
    hp = ExtendedHParams.create_hparams()
    hp.out_dir = str(tmp_path)
    hp.model_name = "m"
    return hp


def test_scalar_writer_falls_back_to_jsonl_and_defers_the_copy(tmp_path):
    hp = _hparams(tmp_path)
    writer = logging_sinks.open_scalar_writer(hp)
    assert writer is logging_sinks.open_scalar_writer(hp)           # one per directory
    sink = logging_sinks.DeferredScalars(writer, flush_every=3)
    for step in range(4):
        sink.add_scalars("Train loss", {"a": torch.tensor(float(step)), "b": torch.tensor(2.0 * step)}, step)
    sink.add_scalars("Validation loss", {"a": torch.tensor(0.5)}, 4)
    sink.flush()
    if isinstance(writer, logging_sinks.JsonlScalarWriter):
        rows = [json.loads(l) for l in open(os.path.join(str(tmp_path), "m", "tensorboard", "scalars.jsonl"))]
        assert [r["step"] for r in rows] == [0, 1, 2, 3, 4]
        assert rows[3] == {"tag": "Train loss", "step": 3, "scalars": {"a": 3.0, "b": 6.0}}
        assert rows[4]["tag"] == "Validation loss" and rows[4]["scalars"] == {"a": 0.5}


def test_scalar_writer_can_be_switched_off(tmp_path):
    hp = _hparams(tmp_path)
    hp.add_hparam("scalar_log_fallback", False)
    hp.add_hparam("tensorboard_dir", os.path.join(str(tmp_path), "tb_off"))
    writer = logging_sinks.open_scalar_writer(hp)
    import importlib.util
    if importlib.util.find_spec("tensorboard") is None:
        assert writer is None
    logging_sinks.DeferredScalars(writer).add_scalars("x", {"a": torch.tensor(1.0)}, 0)      # no-op without a writer


def test_loss_guard_on_the_host_raises_at_once_with_the_reference_messages():
    guard = logging_sinks.DeferredLossCheck("cpu", check_inf=True)
    guard.submit({"mse": torch.tensor(1.0)})
    with pytest.raises(ValueError, match=r"Found NaN in mse loss\."):
        guard.submit({"mse": torch.tensor(float("nan"))})
    with pytest.raises(ValueError, match=r"Found \+/-Inf in mse loss\."):
        guard.submit({"mse": torch.tensor(float("inf"))})
    logging_sinks.DeferredLossCheck("cpu", check_inf=False).submit({"mse": torch.tensor(float("inf"))})
    guard.finish()


def test_memory_probe_without_a_gpu():
    if not torch.cuda.is_available():
        assert logging_sinks.get_gpu_memory_map() == "not available"
    assert logging_sinks.memory_message(False).endswith("GPU: - MB")


@pytest.mark.gpu
def test_loss_guard_on_the_device_raises_one_step_late(gpu):
    dev = torch.device("cuda", 0)
    guard = logging_sinks.DeferredLossCheck(dev, check_inf=True)
    guard.submit({"mse": torch.tensor(1.0, device=dev)})
    guard.submit({"mse": torch.tensor(float("nan"), device=dev)})        # queued, not looked at yet
    with pytest.raises(ValueError, match=r"Found NaN in mse loss\."):
        guard.submit({"mse": torch.tensor(1.0, device=dev)})
    guard = logging_sinks.DeferredLossCheck(dev, check_inf=True)
    guard.submit({"mse": torch.tensor(float("inf"), device=dev)})
    with pytest.raises(ValueError, match=r"Found \+/-Inf in mse loss\."):
        guard.finish()
    usage = logging_sinks.get_gpu_memory_map()
    assert isinstance(usage, dict) and usage[0] > 0
