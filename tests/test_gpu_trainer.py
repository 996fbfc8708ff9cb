"""The reference's trainer tests (test/integration/model_trainers/test_AcousticModelTrainer.py)
re-run against the drop-in AcousticModelTrainer on the GPU stack: same fixture data (packed in
tests/golden/trainer_fixture.npz), same hparams, same assertions -- plus parity of the whole
training trajectory with the losses the reference's CPU run produced."""
import os

import numpy as np
import pytest
import scipy.io.wavfile
import torch

from fixture_dirs import materialise
from idiaptts_amd.src.model_trainers.AcousticModelTrainer import AcousticModelTrainer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fixture(golden_dir, tmp_path_factory):
    root = str(tmp_path_factory.mktemp("trainer_fixture"))
    ids, wdir, qdir, g = materialise(golden_dir, root)
    return root, ids, wdir, qdir, g


def _hparams(root, wdir, name):
    """test_AcousticModelTrainer.py:33-58, on the GPU"""
    hp = AcousticModelTrainer.create_hparams()
    hp.num_questions = 409
    hp.voice = "full"
    hp.out_dir = os.path.join(root, name)
    hp.frame_size_ms = 5
    hp.num_coded_sps = 20
    hp.seed = 1
    hp.epochs = 3
    hp.use_gpu = True
    hp.dataset_num_workers_gpu = 0
    hp.model_type = "RNNDYN-1_RELU_32-1_FC_67"
    hp.batch_size_train = 2
    hp.batch_size_val = 50
    hp.use_saved_learning_rate = True
    hp.optimiser_args["lr"] = 0.001
    hp.model_name = "test_model"
    hp.epochs_per_checkpoint = 2
    hp.world_dir = wdir
    return hp


def _trainer(fixture, hp):
    root, ids, wdir, qdir, g = fixture
    return AcousticModelTrainer(**AcousticModelTrainer.legacy_support_init(
        wdir, qdir, ids, hp.num_questions, hp))


def test_init(gpu, fixture):
    hp = _hparams(fixture[0], fixture[2], "test_init")
    trainer = _trainer(fixture, hp)
    trainer.init(hp)
    nn_dir = os.path.join(hp.out_dir, hp.model_name, hp.networks_dir)
    assert sorted(os.listdir(nn_dir)) == ["config.json", "params_e0"]   # initial checkpoint
    assert next(trainer.model_handler.model.parameters()).is_cuda


def test_train_reproduces_reference_losses(gpu, fixture):
    """test_train (:75-92): seed 1234, 3 epochs, batch size 2, Adam 1e-3, Plateau scheduler.
    The reference asserts that the training loss decreases; with identical initial weights,
    shuffling (same torch RNG stream) and arithmetic the per-epoch losses of the reference's CPU
    run are reproduced by the HIP forward / backward / Adam to float32 accuracy."""
    g = fixture[4]
    hp = _hparams(fixture[0], fixture[2], "test_train")
    hp.seed = 1234
    hp.use_best_as_final_model = False
    trainer = _trainer(fixture, hp)
    trainer.init(hp)
    all_loss, all_loss_train, handler = trainer.train(hp)
    val = all_loss["MSELoss_acoustic_features"]
    train = all_loss_train["MSELoss_acoustic_features"]
    assert train[-1] < train[1 if hp.start_with_test else 0]
    np.testing.assert_allclose(val, g["train_val_losses"], rtol=2e-5)
    np.testing.assert_allclose(train, g["train_train_losses"], rtol=2e-5)
    sd = handler.model.state_dict()
    for k in sd:
        np.testing.assert_allclose(sd[k].cpu().numpy(), g["train_final/" + k], rtol=0, atol=2e-5)
    nn_dir = os.path.join(hp.out_dir, hp.model_name, hp.networks_dir)
    files = set(os.listdir(nn_dir))
    assert {"config.json", "params_e0", "params_e2", "params_e3", "params_best", "optimiser_e3",
            "optimiser_best", "scheduler_e3", "scheduler_best"} <= files
    # resume: the newest checkpoint continues with the saved optimiser / epoch counters
    hp2 = _hparams(fixture[0], fixture[2], "test_train")
    hp2.seed = 1234
    hp2.load_checkpoint_epoch = 3
    hp2.epochs = 1
    trainer2 = _trainer(fixture, hp2)
    trainer2.init(hp2)
    assert (trainer2.total_epoch, trainer2.total_steps) == (3, trainer.total_steps)
    _, train2, _ = trainer2.train(hp2)
    assert trainer2.total_epoch == 4
    assert train2["MSELoss_acoustic_features"][-1] < train[-1]


def test_benchmark(gpu, fixture):
    """test_benchmark (:94-106): the untrained seed-1 model scores (8.616, 78.4, 0.609, 37.352)."""
    hp = _hparams(fixture[0], fixture[2], "test_benchmark")
    trainer = _trainer(fixture, hp)
    trainer.init(hp)
    scores = trainer.benchmark(hp)
    np.testing.assert_almost_equal((8.616, 78.4, 0.609, 37.352),
                                   scores["pred_acoustic_features"], 3)


def test_synth_wav_and_copy_synth(gpu, fixture):
    """test_synth_wav / test_copy_synth (:131-178): one wav per id in
    <out_dir>/<model>/synth/e<epoch>, named <id>_<n><sp_type>_WORLD.wav (copy synthesis:
    <id>_ref_... in hparams.synth_dir)."""
    root, ids = fixture[0], fixture[1]
    hp = _hparams(root, fixture[2], "test_synth")
    hp.synth_fs = 16000
    trainer = _trainer(fixture, hp)
    trainer.init(hp)
    outputs, outputs_post = trainer.synth(hp, ids[:2])
    assert set(outputs) == set(ids[:2])
    synth_dir = os.path.join(hp.out_dir, hp.model_name, "synth", "e0")
    for i in ids[:2]:
        assert outputs[i]["pred_acoustic_features"].shape[1] == 67
        post = outputs_post[i]["pred_acoustic_features"]
        assert post.shape == (len(outputs[i]["pred_acoustic_features"]), 23)   # static after MLPG
        fs, wav = scipy.io.wavfile.read(os.path.join(synth_dir, i + "_20mcep_WORLD.wav"))
        assert fs == 16000 and len(wav) == len(post) * 80      # pyworld.synthesize length
    hp.synth_dir = os.path.join(hp.out_dir, "copy")
    os.makedirs(hp.synth_dir, exist_ok=True)
    trainer.copy_synth(hp, ids[:2])
    for i in ids[:2]:
        fs, wav = scipy.io.wavfile.read(os.path.join(hp.synth_dir, i + "_ref_20mcep_WORLD.wav"))
        assert fs == 16000 and np.abs(wav).max() > 1000          # real speech comes back


def test_train_with_hbm_resident_shards_reproduces_reference_losses(gpu, fixture):
    """hparams.resident_dataset: readers run once, the normalised frames live
    in HBM as packed shards, the epochs run the flat feed-forward step on gathered mini-batches.
    Same shuffling (same RNG stream), same arithmetic: the reference's per-epoch losses again."""
    g = fixture[4]
    hp = _hparams(fixture[0], fixture[2], "test_train_resident")
    hp.seed = 1234
    hp.use_best_as_final_model = False
    hp.resident_dataset = True
    trainer = _trainer(fixture, hp)
    trainer.init(hp)
    all_loss, all_loss_train, handler = trainer.train(hp)
    np.testing.assert_allclose(all_loss["MSELoss_acoustic_features"], g["train_val_losses"],
                               rtol=2e-5)
    np.testing.assert_allclose(all_loss_train["MSELoss_acoustic_features"],
                               g["train_train_losses"], rtol=2e-5)
    sd = handler.model.state_dict()          # flat parameters were written back to the modules
    for k in sd:
        np.testing.assert_allclose(sd[k].cpu().numpy(), g["train_final/" + k], rtol=0, atol=2e-5)
    shard = handler._resident["shards"]["train"]
    assert shard.x.is_cuda and shard.x.shape[1] == 412 and len(shard) == 7
    # the saved checkpoint holds the trained weights and an optimiser state a module-path run can
    # continue from
    nn_dir = os.path.join(hp.out_dir, hp.model_name, hp.networks_dir)
    ck = torch.load(os.path.join(nn_dir, "params_e3"), map_location="cpu", weights_only=False)
    for k in sd:
        assert torch.equal(ck["params"][k], sd[k].cpu())
    opt = torch.load(os.path.join(nn_dir, "optimiser_e3"), map_location="cpu", weights_only=False)
    assert len(opt["params"]["state"]) == 4 and opt["params"]["state"][0]["step"] == 12   # 4 batches x 3 epochs
    scores = trainer.benchmark(hp)["pred_acoustic_features"]
    assert np.isfinite(scores).all()


def test_duration_model_trainer(gpu, golden_dir, tmp_path):
    """BASELINE config 4, the recipe of the reference's (commented-out) test_DurationModelTrainer:
    one-hot phonemes -> 5 state durations, RNNDYN-1_RELU_32-1_FC_5, then a BiLSTM variant; the
    loss decreases, benchmark scores are finite, forward returns whole-frame durations in HTK
    units."""
    from fixture_dirs import materialise_duration
    from idiaptts_amd.src.model_trainers.DurationModelTrainer import DurationModelTrainer
    root = str(tmp_path)
    ids, g = materialise_duration(golden_dir, root)
    for model_type in ("RNNDYN-1_RELU_32-1_FC_5", "RNNDYN-1_BiLSTM_16-1_FC_5"):
        hp = DurationModelTrainer.create_hparams()
        hp.out_dir = os.path.join(root, "out_" + model_type[7:13])
        hp.seed = 1234
        hp.epochs = 6
        hp.use_gpu = True
        hp.dataset_num_workers_gpu = 0
        hp.model_type = model_type
        hp.batch_size_train = 4
        hp.batch_size_val = 64
        hp.optimiser_args["lr"] = 0.01
        hp.model_name = "test_model.nn"
        hp.use_best_as_final_model = False
        trainer = DurationModelTrainer(**DurationModelTrainer.legacy_support_init(
            os.path.join(root, "labels", "label_state_align"), os.path.join(root, "dur"), ids,
            os.path.join(root, "labels", "mono_phone.list"), hp))
        trainer.init(hp)
        assert trainer.model_handler.model.model.config.in_dim == 59
        _, train_loss, _ = trainer.train(hp)
        train_loss = train_loss["MSELoss_durations"]
        assert train_loss[-1] < train_loss[0]
        rmse, pearson = trainer.benchmark(hp)["pred_durations"]
        assert np.isfinite(rmse) and rmse > 0 and pearson.shape == (5,)
        out, post = trainer.forward(hp, ids[:3])
        for i in ids[:3]:
            assert post[i].shape == g["dur/" + i].shape and post[i].dtype == np.int64
            assert (post[i] % hp.min_phoneme_length == 0).all() and (post[i] >= 0).all()
            assert out[i]["pred_durations"].shape == g["dur/" + i].shape


def test_ema_and_gradient_clipping_module_path_equals_resident_path(gpu, fixture):
    """hparams.ema_decay / grad_clip_* (reference handler :810-831): the module path (HipAdam's
    fused step on its flat arena) and the resident path (FlatFFModel's fused step) are separate
    plumbing over the same kernels (pinned against torch in tests/test_gpu_optim.py) -- they must
    produce the same losses; validation runs on the averaged parameters in both."""
    runs = []
    for resident in (False, True):
        hp = _hparams(fixture[0], fixture[2], "test_ema_clip_{}".format(int(resident)))
        hp.seed = 1234
        hp.use_best_as_final_model = False
        hp.resident_dataset = resident
        hp.ema_decay = 0.8
        hp.grad_clip_norm_type = 2
        hp.grad_clip_max_norm = 0.05
        hp.grad_clip_thresh = 0.01
        trainer = _trainer(fixture, hp)
        trainer.init(hp)
        val, train, handler = trainer.train(hp)
        assert handler.ema is not None and handler.ema.fused
        ema_sd = {k: v.detach().cpu().numpy().copy() for k, v in handler.ema.model.state_dict().items()}
        live_sd = {k: v.detach().cpu().numpy().copy() for k, v in handler.model.state_dict().items()}
        runs.append((val["MSELoss_acoustic_features"], train["MSELoss_acoustic_features"],
                     ema_sd, live_sd))
    (v0, t0, e0, l0), (v1, t1, e1, l1) = runs
    np.testing.assert_allclose(v0, v1, rtol=2e-5)
    np.testing.assert_allclose(t0, t1, rtol=2e-5)
    for k in e0:
        np.testing.assert_allclose(e0[k], e1[k], rtol=0, atol=2e-6)
        np.testing.assert_allclose(l0[k], l1[k], rtol=0, atol=2e-6)
        assert np.abs(e0[k] - l0[k]).max() > 0          # the averaged model lags the live one
    g = fixture[4]
    assert np.abs(np.asarray(t0[1:]) - np.asarray(g["train_train_losses"][1:])).max() > 1e-6   # clipping acted


def test_tts_chain_labels_to_waveform(gpu, fixture, golden_dir, tmp_path):
    """TTSModel.run_DM_AM behind the Festival front end (reference TTSModel.py:100-165, BASELINE
    config 4 end to end): mono labels -> duration model -> state-aligned full labels -> question
    labels -> acoustic model -> MLPG -> WORLD.  Both models are trained for a few epochs on the
    fixtures first; the labels of the 'new' utterances are the fixture labels stripped of their
    timing."""
    import re
    from fixture_dirs import materialise_duration
    from idiaptts_amd.src.model_trainers.DurationModelTrainer import DurationModelTrainer
    from idiaptts_amd.src.TTSModel import TTSModel
    root = str(tmp_path)
    ids, g = materialise_duration(golden_dir, root)
    # ---- duration model
    hp = DurationModelTrainer.create_hparams()
    hp.out_dir, hp.seed, hp.epochs, hp.use_gpu = os.path.join(root, "dm"), 1, 8, True
    hp.dataset_num_workers_gpu = 0
    hp.model_type, hp.model_name = "RNNDYN-1_RELU_32-1_FC_5", "dm"
    hp.batch_size_train, hp.optimiser_args["lr"] = 4, 0.01
    dm = DurationModelTrainer(**DurationModelTrainer.legacy_support_init(
        os.path.join(root, "labels", "label_state_align"), os.path.join(root, "dur"), ids,
        os.path.join(root, "labels", "mono_phone.list"), hp))
    dm.init(hp)
    dm.train(hp)
    # ---- acoustic model (the fixture recipe of test_train)
    hpa = _hparams(fixture[0], fixture[2], "tts_am")
    hpa.seed = 1234
    am = _trainer(fixture, hpa)
    am.init(hpa)
    am.train(hpa)
    # ---- "front end output": mono + full labels without state alignment for two utterances
    work = os.path.join(root, "work")
    new_ids = ids[:2]
    for sub in ("mono", "full"):
        os.makedirs(os.path.join(work, "labels", sub))
    for i in new_ids:
        with open(os.path.join(root, "labels", "label_state_align", i + ".lab")) as f:
            lines = [l.split() for l in f if l.strip()]
        phones = lines[::5]                                           # five state lines per phone
        with open(os.path.join(work, "labels", "full", i + ".lab"), "w") as f:
            f.write("\n".join("0 0 " + re.sub(r"\[\d+\]$", "", p[2]) for p in phones))
        with open(os.path.join(work, "labels", "mono", i + ".lab"), "w") as f:
            f.write("\n".join("0 0 " + re.search(r"-(.+?)\+", p[2]).group(1) for p in phones))
    # ---- the chain
    hpt = TTSModel.create_hparams()
    hpt.use_gpu, hpt.dataset_num_workers_gpu = True, 0
    hpt.num_coded_sps, hpt.num_questions, hpt.frame_size_ms = 20, 409, 5
    hpt.duration_labels_dir = os.path.join(root, "dur")
    hpt.file_symbol_dict = os.path.join(root, "labels", "mono_phone.list")
    hpt.duration_model = os.path.join(hp.out_dir, hp.model_name, hp.networks_dir)
    hpt.acoustic_model = os.path.join(hpa.out_dir, hpa.model_name, hpa.networks_dir)
    hpt.question_file = os.path.join(golden_dir, "questions-en-radio_dnn_400.hed")
    hpt.question_labels_norm_file = os.path.join(fixture[3], "min-max.bin")
    hpt.world_features_dir = fixture[2]
    hpt.synth_dir = os.path.join(root, "tts_out")
    durations, features = TTSModel.run_DM_AM_on_labels(hpt, work, new_ids)
    for i in new_ids:
        n_frames = int(durations[i].sum() // hpt.min_phoneme_length)
        assert durations[i].shape == g["dur/" + i].shape and n_frames > 50
        with open(os.path.join(work, "labels", "label_state_align", i + ".lab")) as f:
            last = f.read().split("\n")[-2].split("\t")
        assert int(last[1]) == int(durations[i].sum()) and last[2].endswith("[6]")
        cmp_ = features[i]["pred_acoustic_features"]
        assert cmp_.shape[0] == n_frames and np.isfinite(cmp_).all()
        wavs = [f for f in os.listdir(hpt.synth_dir) if f.startswith(i) and f.endswith(".wav")]
        assert len(wavs) == 1
        from scipy.io import wavfile
        fs, w = wavfile.read(os.path.join(hpt.synth_dir, wavs[0]))
        assert fs == 16000 and len(w) == n_frames * 80 and np.abs(w).max() > 0


def test_async_checkpoints_equal_synchronous_ones(gpu, fixture):
    """hparams.async_checkpoint: the background writer produces the same files with the same
    contents as the synchronous torch.save calls, and resuming from them works."""
    outs = {}
    for mode in (False, True):
        hp = _hparams(fixture[0], fixture[2], "test_async_{}".format(int(mode)))
        hp.seed = 1234
        hp.use_best_as_final_model = True          # exercises a checkpoint read inside train()
        hp.async_checkpoint = mode
        trainer = _trainer(fixture, hp)
        trainer.init(hp)
        val, train, handler = trainer.train(hp)
        nn_dir = os.path.join(hp.out_dir, hp.model_name, hp.networks_dir)
        outs[mode] = (nn_dir, val["MSELoss_acoustic_features"])
    (d0, v0), (d1, v1) = outs[False], outs[True]
    np.testing.assert_allclose(v0, v1, rtol=0, atol=0)
    assert sorted(os.listdir(d0)) == sorted(os.listdir(d1))
    for f in os.listdir(d0):
        if f == "config.json" or f.startswith("scheduler"):
            continue
        a = torch.load(os.path.join(d0, f), map_location="cpu", weights_only=False)
        b = torch.load(os.path.join(d1, f), map_location="cpu", weights_only=False)
        assert a["epoch"] == b["epoch"]
        pa, pb = a["params"], b["params"]
        if "state" in pa:                          # optimiser file
            for k in pa["state"]:
                assert torch.equal(pa["state"][k]["exp_avg"], pb["state"][k]["exp_avg"])
        else:
            for k in pa:
                assert torch.equal(pa[k], pb[k])
