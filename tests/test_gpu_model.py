"""GPU parity of the drop-in model stack (RNNDyn + NamedForwardWrapper + NamedLoss on the HIP
kernels) against outputs captured from the reference's own modules on torch CPU
(tests/golden/model_forward.npz, made by tests/golden/make_golden.py)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _wrapped(model_type, in_dim):
    from idiaptts_amd.src.neural_networks.pytorch.models import rnn_dyn
    from idiaptts_amd.src.neural_networks.pytorch.models.NamedForwardWrapper import \
        NamedForwardWrapper
    hp = types.SimpleNamespace(model_type=model_type, batch_first=False, dropout=0.0)
    return NamedForwardWrapper.Config(rnn_dyn.convert_legacy_to_config((in_dim,), hp),
                                      input_names=["questions"], batch_first=False,
                                      name="AcousticModel",
                                      output_names=["pred_acoustic_features"])


def test_fixture_checkpoint_forward_matches_reference(gpu, golden_dir):
    g = np.load(os.path.join(golden_dir, "model_forward.npz"))
    model = _wrapped("RNNDYN-1_RELU_32-1_FC_67", 409).create_model()
    model.load_state_dict({k[len("a_sd_"):]: torch.from_numpy(g[k]) for k in g.files
                           if k.startswith("a_sd_")})
    model = model.to(gpu)
    q = torch.from_numpy(g["a_questions"])[:, None, :].to(gpu)
    data, lengths = {"questions": q}, {"questions": torch.tensor([q.shape[0]])}
    model.init_hidden(1)
    with torch.no_grad():
        model(data, lengths, {"questions": q.shape[0]})
    pred = data["pred_acoustic_features"].cpu().numpy()
    assert pred.shape == g["a_pred"].shape
    assert np.abs(pred - g["a_pred"]).max() < 2e-5      # fp32: tolerance 2e-5 abs on O(1) outputs


def test_bilstm_stack_forward_loss_grads_match_reference(gpu, golden_dir):
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler
    from idiaptts_amd.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    g = np.load(os.path.join(golden_dir, "model_forward.npz"))
    model = _wrapped("RNNDYN-2_TANH_24-2_BiLSTM_16-1_FC_7", 11).create_model()
    model.load_state_dict({k[len("b_sd_"):]: torch.from_numpy(g[k]) for k in g.files
                           if k.startswith("b_sd_")})
    model = model.to(gpu)
    lens = torch.from_numpy(g["b_len"])
    data = {"questions": torch.from_numpy(g["b_x"]).to(gpu),
            "acoustic_features": torch.from_numpy(g["b_tgt"]).to(gpu),
            "acoustic_features_mask": Handler.sequence_mask(lens, 9, batch_first=False).to(gpu)}
    lengths = {"questions": lens, "acoustic_features": lens, "acoustic_features_mask": lens}
    model.init_hidden(3)
    model(data, lengths, {"questions": 9})
    pred = data["pred_acoustic_features"]
    assert np.abs(pred.detach().cpu().numpy() - g["b_pred"]).max() < 2e-5
    loss_mod = NamedLoss.Config(name="MSELoss_acoustic_features", type_="MSELoss",
                                seq_mask="acoustic_features_mask",
                                input_names=["acoustic_features", "pred_acoustic_features"],
                                batch_first=False).create_loss()
    loss = list(loss_mod(data, lengths, step=1).values())[0]
    assert abs(loss.item() - float(g["b_loss"])) < 1e-5 * max(1.0, abs(float(g["b_loss"])))
    loss.backward()
    for k, p in model.named_parameters():
        ref = g["b_grad_" + k]
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err < 1e-4 * max(1e-2, np.abs(ref).max()), (k, err)


def test_handler_trains_and_checkpoints(gpu, tmp_path):
    """3 steps of the handler loop: loss decreases (reference test_AcousticModelTrainer.py:76-92
    in spirit), checkpoint round trip with the reference's file names."""
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler
    from idiaptts_amd.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    torch.manual_seed(0)
    h = Handler()
    h.create_model(_wrapped("RNNDYN-1_TANH_32-1_BiLSTM_16-1_FC_5", 12))
    h.set_optimiser("Adam", lr=1e-2)
    h.set_losses([NamedLoss.Config(name="MSELoss_acoustic_features", type_="MSELoss",
                                   seq_mask="acoustic_features_mask",
                                   input_names=["acoustic_features", "pred_acoustic_features"],
                                   batch_first=False)])
    rng = np.random.default_rng(1)
    batch = [{"questions": rng.normal(size=(t, 12)).astype(np.float32),
              "acoustic_features": rng.normal(size=(t, 5)).astype(np.float32)} for t in (30, 21, 17)]
    losses = []
    for step in range(12):
        data, lengths = Handler.prepare_batch(batch, batch_first=False,
                                              mask_keys=("acoustic_features",))
        ld, _ = h.process_batch(data, lengths, step, training=True)
        losses.append(ld["MSELoss_acoustic_features"])
    assert losses[-1] < losses[0]
    h.save_checkpoint(str(tmp_path), epoch=1, step=12)
    assert os.path.isfile(os.path.join(str(tmp_path), "params_e1"))
    assert os.path.isfile(os.path.join(str(tmp_path), "optimiser_e1"))
    assert os.path.isfile(os.path.join(str(tmp_path), "config.json"))
    from idiaptts_amd.src.ExtendedHParams import ExtendedHParams
    hp = ExtendedHParams.create_hparams()
    hp.use_gpu = True
    hp.optimiser_args["lr"] = 1e-2
    h2 = Handler()                       # the model is rebuilt from config.json
    best_loss, epoch, step = h2.load_checkpoint(hp, str(tmp_path), epoch=1)
    assert (epoch, step) == (1, 12) and np.isinf(best_loss)
    st1, st2 = h.optimiser.state_dict()["state"], h2.optimiser.state_dict()["state"]
    assert st1.keys() == st2.keys() and all(
        torch.equal(st1[k]["exp_avg"].cpu(), st2[k]["exp_avg"].cpu()) for k in st1)
    for (k, a), (_, b) in zip(h.model.state_dict().items(), h2.model.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k


def test_benchmark_known_answer_end_to_end(gpu, golden_dir):
    """The reference's pinned benchmark (test_AcousticModelTrainer.py:94-106: MCD 8.616, F0-RMSE
    78.4, VDE 0.609, BAP 37.352 for an untrained seed-1 RNNDYN-1_RELU_32-1_FC_67) recomputed on
    the GPU stack: HIP forward of the reference's weights on its normalised test-utterance input,
    de-normalisation, HIP MLPG per stream (_postprocess_world), objective scores."""
    from idiaptts_amd.src.Metrics import Metrics
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    g = np.load(os.path.join(golden_dir, "benchmark_e2e.npz"))
    model = _wrapped("RNNDYN-1_RELU_32-1_FC_67", 409).create_model()
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")})
    model = model.to(gpu)
    q = torch.from_numpy(g["questions_norm"]).to(gpu)             # [T, 1, 409] time-major
    T = q.shape[0]
    data, lengths = {"questions": q}, {"questions": torch.tensor([T])}
    model.init_hidden(1)
    with torch.no_grad():
        model(data, lengths, {"questions": T})
    pred = data["pred_acoustic_features"][:, 0].cpu().numpy()
    denorm = pred * g["out_std"] + g["out_mean"]                  # NpzDataReader.postprocess_sample
    assert np.abs(denorm - g["denormalised"]).max() < 1e-4
    gen = WorldFeatLabelGen(None, add_deltas=True, num_coded_sps=20, num_bap=1)
    gen.covs = [g["cov_0"], g["cov_1"], None, g["cov_3"]]
    post = gen._postprocess_world(np.array(denorm, dtype=np.float64))
    out = WorldFeatLabelGen.convert_to_world_features(post, contains_deltas=False,
                                                      num_coded_sps=20, num_bap=1)
    org = WorldFeatLabelGen.convert_to_world_features(g["original"], contains_deltas=True,
                                                      num_coded_sps=20, num_bap=1)
    scores = Metrics.get_metrics(org, out)
    np.testing.assert_almost_equal((8.616, 78.4, 0.609, 37.352), scores, 3)
    assert np.allclose(scores, g["scores"], rtol=1e-5)


def test_embedding_group_matches_reference(gpu, golden_dir):
    """RNNDyn with an embedding group (rnn_dyn/RNNDyn.py:39-49, 88-134): the index rides in the
    last input column, its vector is concatenated in front of groups 0 and 2.  Output and all
    gradients against the reference's own module (tests/golden/make_golden.py --embedding)."""
    from idiaptts_amd.src.neural_networks.pytorch.models.rnn_dyn.Config import Config
    g = np.load(os.path.join(golden_dir, "model_embedding.npz"))
    cfg = Config(in_dim=6, batch_first=False, layer_configs=[
        Config.LayerConfig(layer_type="Linear", out_dim=8, num_layers=1, nonlin="tanh"),
        Config.LayerConfig(layer_type="GRU", out_dim=16, num_layers=1, bidirectional=True),
        Config.LayerConfig(layer_type="Linear", out_dim=4, num_layers=1)],
        emb_configs=[Config.EmbeddingConfig(embedding_dim=3, name="emb_speaker", num_embedding=4,
                                            affected_layer_group_indices=(0, 2))])
    model = cfg.create_model()
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    assert set(sd) == set(model.state_dict())
    model.load_state_dict(sd)
    model.to(gpu)
    assert model.get_group_out_dim(1) == 32 and model[0].out_dim == 8
    x = torch.from_numpy(np.concatenate([g["x"], g["idx"]], axis=2)).to(gpu)
    lens = torch.from_numpy(g["len"])
    model.init_hidden(3)
    out, _ = model(x, seq_lengths_input=lens, max_length_inputs=7)
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() < 2e-6
    (out * torch.from_numpy(g["w"]).to(gpu)).sum().backward()
    for name, p in model.named_parameters():
        ref = g["grad_" + name]
        assert np.abs(p.grad.cpu().numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), name
    # the same through explicit embedding inputs
    model.zero_grad()
    model.init_hidden(3)
    out2, _ = model(torch.from_numpy(g["x"]).to(gpu), torch.from_numpy(g["idx"]).to(gpu),
                    seq_lengths_input=lens, max_length_inputs=7)
    assert torch.equal(out2, out)
