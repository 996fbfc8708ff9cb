"""Host logic of the trainer drop-in (no GPU): id partition, parameter initialisation, readers,
length matching and normalisation against values captured from the reference
(tests/golden/make_golden.py --trainer / --kat), config.json interchange, hparams surface."""
import json
import os

import numpy as np
import pytest
import torch

from fixture_dirs import materialise
from idiaptts_amd.src.ExtendedHParams import ExtendedHParams
from idiaptts_amd.src.model_trainers.AcousticModelTrainer import AcousticModelTrainer
from idiaptts_amd.src.neural_networks.pytorch import config_json


def _hparams(root, wdir, seed=1):
    """reference test_AcousticModelTrainer.py:33-58"""
    hp = AcousticModelTrainer.create_hparams()
    hp.num_questions = 409
    hp.voice = "full"
    hp.out_dir = os.path.join(root, "out")
    hp.frame_size_ms = 5
    hp.num_coded_sps = 20
    hp.seed = seed
    hp.epochs = 3
    hp.use_gpu = False
    hp.model_type = "RNNDYN-1_RELU_32-1_FC_67"
    hp.batch_size_train = 2
    hp.batch_size_val = 50
    hp.use_saved_learning_rate = True
    hp.optimiser_args["lr"] = 0.001
    hp.model_name = "test_model"
    hp.epochs_per_checkpoint = 2
    hp.world_dir = wdir
    return hp


@pytest.fixture(scope="module")
def fixture(golden_dir, tmp_path_factory):
    root = str(tmp_path_factory.mktemp("trainer_fixture"))
    ids, wdir, qdir, g = materialise(golden_dir, root)
    return root, ids, wdir, qdir, g


@pytest.mark.parametrize("seed,tag", [(1, "seed1"), (1234, "train")])
def test_id_partition_and_initial_weights_match_reference(fixture, seed, tag):
    root, ids, wdir, qdir, g = fixture
    hp = _hparams(root, wdir, seed)
    trainer = AcousticModelTrainer(**AcousticModelTrainer.legacy_support_init(
        wdir, qdir, ids, hp.num_questions, hp))
    assert trainer.id_list_train == [str(i) for i in g[tag + "_ids_train"]]
    assert trainer.id_list_val == [str(i) for i in g[tag + "_ids_val"]]
    assert trainer.id_list_test == [str(i) for i in g[tag + "_ids_test"]]
    hp.epochs = 0                                   # no initial checkpoint
    trainer.init(hp)
    sd = trainer.model_handler.model.state_dict()
    keys = [k[len(tag) + 6:] for k in g.files if k.startswith(tag + "_init/")]
    assert sorted(keys) == sorted(sd.keys())
    for k in keys:
        assert np.array_equal(sd[k].numpy(), g[tag + "_init/" + k]), k
    assert hp.scheduler_type == "Plateau"           # AcousticModelTrainer's default scheduler


def test_dataset_item_matches_reference_inference_input(fixture, golden_dir):
    root, ids, wdir, qdir, g = fixture
    e2e = np.load(os.path.join(golden_dir, "benchmark_e2e.npz"))
    hp = _hparams(root, wdir, 1)
    hp.epochs = 0
    trainer = AcousticModelTrainer(**AcousticModelTrainer.legacy_support_init(
        wdir, qdir, ids, hp.num_questions, hp))
    trainer.init(hp)
    assert trainer.id_list_test == [str(e2e["test_id"])]
    item, dataset = trainer.dataset_test[0]
    # question labels are trimmed symmetrically to the (shorter) acoustic features, both
    # normalised exactly like the reference's readers
    assert item["questions"].shape == (1132, 409) and item["questions"].dtype == np.float32
    assert np.array_equal(item["questions"], e2e["questions_norm"][:, 0])
    wreader = trainer.datareaders["acoustic_features"]
    assert np.array_equal(np.asarray(wreader.norm_params[0], dtype=np.float64), e2e["out_mean"])
    assert np.array_equal(np.asarray(wreader.norm_params[1], dtype=np.float64), e2e["out_std"])
    for i in (0, 1, 3):
        assert np.array_equal(np.asarray(wreader.covs[i], dtype=np.float64), e2e["cov_%d" % i])
    assert item["acoustic_features"].shape == (1132, 67)
    raw = wreader.load(trainer.id_list_test[0])
    assert raw.shape[0] == 1137 - 5 or raw.shape[0] >= 1132
    org = trainer.get_output_dict(trainer.id_list_test, hp)[trainer.id_list_test[0]]
    assert np.array_equal(org, e2e["original"])
    # collate: time-major padding, mask for the reader that asks for one
    batch = [trainer.dataset_train[i] for i in range(3)]
    data, lengths = trainer.model_handler.prepare_batch(batch, batch_first=False)
    T = int(lengths["questions"].max())
    assert data["questions"].shape == (T, 3, 409) and data["acoustic_features"].shape == (T, 3, 67)
    assert data["acoustic_features_mask"].shape == (T, 3, 1) and "questions_mask" not in data
    assert torch.equal(lengths["acoustic_features_mask"], lengths["acoustic_features"])
    assert torch.equal(data["acoustic_features_mask"].sum(dim=(0, 2)).long(),
                       lengths["acoustic_features"])
    assert data["_id_list"] == trainer.id_list_train[:3]


def test_config_json_interchange_with_reference_layout(golden_dir):
    path = os.path.join(golden_dir, "model_in409_out67_config.json")
    with open(path) as f:
        text = f.read()
    cfg = config_json.decode(text)                  # a file written by the reference
    model = cfg.create_model()
    assert [tuple(p.shape) for p in model.parameters()] == [(32, 409), (32,), (67, 32), (67,)]
    assert list(model.state_dict().keys()) == ["model.1.module.0.weight", "model.1.module.0.bias",
                                               "model.2.module.0.weight", "model.2.module.0.bias"]
    assert json.loads(config_json.encode(cfg)) == json.loads(text)   # and back, tag for tag


def test_hparams_container_surface():
    hp = ExtendedHParams.create_hparams("epochs=7,model_name=abc,optimiser_type=SGD")
    assert hp.epochs == 7 and hp.model_name == "abc" and hp.optimiser_type == "SGD"
    assert hp.batch_size_val == 48 and hp.networks_dir == "nn" and hp.synth_vocoder == "WORLD"
    assert not hp.has_value("model_path") and hp.get_value("model_path", "x") == "x"
    hp.add_hparams(new_value=3)
    hp.setattr_no_type_check("backprop_loss_names", ["a"])
    assert hp.new_value == 3 and hp.has_value("backprop_loss_names")
    with pytest.raises(ValueError):
        hp.add_hparam("epochs", 1)
    assert "epochs=7" in hp.get_debug_string()
    ahp = AcousticModelTrainer.create_hparams()
    assert ahp.add_deltas and ahp.num_coded_sps == 60 and len(ahp.metrics) == 4


def test_split_batch_matches_reference_semantics():
    from idiaptts_amd.src.model_trainers.ModularTrainer import ModularTrainer
    x = np.arange(2 * 3 * 4, dtype=np.float32).reshape(4, 3, 2)      # [T, B, D]
    out = ModularTrainer.split_batch({"a": x, "ids": ["u", "v", "w"]},
                                     {"a": np.array([4, 2, 3])}, batch_first=False)
    assert [o.shape for o in out["a"]] == [(4, 2), (2, 2), (3, 2)]
    assert np.array_equal(out["a"][1], x[:2, 1]) and out["ids"] == ["u", "v", "w"]
    h = (np.zeros((3, 5)), None)
    parts = ModularTrainer._split_return_values(h, None, batch_first=True)
    assert len(parts) == 3 and parts[0][1] is None and parts[0][0].shape == (5,)
