#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ (run in the BUILD container only;
/root/reference does not exist on the GPU box and nothing here travels except the outputs).

1. Copies DATA fixtures the reference's own tests hold (two short 16 kHz wavs and their golden
   .cmp feature files, test/integration/fixtures/{database/wav,WORLD/cmp_mcep20}).
2. Imports the reference's Python (stub harness of SURVEY.md Appendix A: MagicMock modules for
   the absent third-party packages) and records input/output pairs of its own logic:
   interpolate_lin, compute_deltas, convert_to_world_features, trim_to_shortest,
   PyTorchDatareadersDataset._trim_datareader_output, sequence_mask, NamedLoss(mean_per_frame).
3. Runs the reference's AcousticModelTrainer.benchmark known-answer test
   (test/integration/model_trainers/test_AcousticModelTrainer.py:94-106) with the C oracle's MLPG
   standing in for bandmat, asserts the pinned scores (8.616, 78.4, 0.609, 37.352) and stores
   the MLPG inputs/outputs of that run.
"""
import base64
import logging
import os
import pickle
import shutil
import sys
import types
import warnings
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
FIX = os.path.join(REF, "test", "integration", "fixtures")
sys.path.insert(0, ROOT)


def install_stub_harness():
    warnings.simplefilter("ignore")
    logging.raiseExceptions = False
    os.environ["IDIAPTTS_ROOT"] = os.path.join(REF, "idiaptts")
    sys.path.insert(0, REF)

    class Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            m = MagicMock(name=self.__name__ + "." + k)
            setattr(self, k, m)
            return m

    for n in ["git", "git.exc", "pyworld", "pyworld.pyworld", "pysptk", "pysptk.util", "bandmat",
              "bandmat.linalg", "librosa", "librosa.display", "librosa.feature", "nnmnkwii",
              "nnmnkwii.postfilters", "soundfile", "pydub", "pydub.utils", "wavenet_vocoder",
              "wavenet_vocoder.util", "wavenet_vocoder.modules", "wavenet_vocoder.wavenet",
              "wavenet_vocoder.mixture", "textgrid", "torchinfo", "tensorboard",
              "numpy.lib.arraysetops"]:
        m = Stub(n)
        m.__path__ = []
        sys.modules[n] = m

    class _E(Exception):
        pass
    sys.modules["git.exc"].InvalidGitRepositoryError = _E
    sys.modules["git"].exc = sys.modules["git.exc"]

    def _repo(*a, **k):
        raise _E()
    sys.modules["git"].Repo = _repo
    np.str = str
    np.int = int
    np.float = float
    jp = types.ModuleType("jsonpickle")
    jp.encode = lambda o, indent=None: base64.b64encode(pickle.dumps(o)).decode()
    jp.decode = lambda s: pickle.loads(base64.b64decode(s))
    sys.modules["jsonpickle"] = jp


def copy_data_fixtures():
    for name in ["LJ001-0002", "LJ001-0008"]:
        shutil.copyfile(os.path.join(FIX, "database", "wav", name + ".wav"),
                        os.path.join(HERE, name + ".wav"))
        shutil.copyfile(os.path.join(FIX, "WORLD", "cmp_mcep20", name + ".cmp"),
                        os.path.join(HERE, name + ".cmp"))
        os.chmod(os.path.join(HERE, name + ".wav"), 0o644)
        os.chmod(os.path.join(HERE, name + ".cmp"), 0o644)


def capture_host_logic():
    import torch
    from idiaptts.misc.utils import compute_deltas, interpolate_lin
    from idiaptts.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    from idiaptts.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler

    rng = np.random.default_rng(2024)
    out = {}
    # interpolate_lin: Appendix-C cases + random float32 / float64 contours
    cases = [[0, 0, 5, 5.2, 0, 0, 5.8, 0], [5, 0, 0, 0, 6, 0], [0, 0, 5, 0, 0, 6],
             [0, 0, 5, 0, 0, 6, 6], [5, 0, 6], [0, 5], [0, 0, 0], [4.5]]
    arrs = [np.array(c, dtype=np.float32) for c in cases]
    for t in range(40):
        n = int(rng.integers(1, 60))
        x = (rng.uniform(3, 6, size=n) * (rng.uniform(size=n) < rng.uniform(0.1, 0.9)))
        arrs.append(x.astype(np.float32 if t % 4 else np.float64))
    for i, a in enumerate(arrs):
        ip, vuv = interpolate_lin(a)
        out["il_in_%d" % i] = a
        out["il_ip_%d" % i] = ip
        out["il_vuv_%d" % i] = vuv
    out["il_count"] = np.array(len(arrs))
    # compute_deltas
    for i, shape in enumerate([(4, 2), (2, 3), (57, 5), (300, 1)]):
        x = rng.normal(size=shape).astype(np.float32)
        out["cd_in_%d" % i] = x
        out["cd_out_%d" % i] = compute_deltas(x)
    out["cd_count"] = np.array(4)
    # convert_to_world_features with / without deltas (20 coded sps, 1 bap)
    for i, (deltas, ncs, nb) in enumerate([(False, 20, 1), (True, 20, 1), (True, 60, 1),
                                           (False, 60, 5)]):
        width = (ncs + 1 + nb) * (3 if deltas else 1) + 1
        s = rng.uniform(-1, 1.5, size=(13, width)).astype(np.float32)
        c, l, v, b = WorldFeatLabelGen.convert_to_world_features(s, deltas, ncs, nb)
        out["cw_in_%d" % i] = s
        out["cw_meta_%d" % i] = np.array([int(deltas), ncs, nb])
        out["cw_sp_%d" % i] = c
        out["cw_lf0_%d" % i] = l
        out["cw_vuv_%d" % i] = v
        out["cw_bap_%d" % i] = b
    out["cw_count"] = np.array(4)
    # trim_to_shortest: symmetric trimming (front = diff//2)
    feats = [np.arange(20, dtype=np.float32).reshape(10, 2), np.arange(7, dtype=np.float32)[:, None],
             None, np.arange(8, dtype=np.float32)[:, None]]
    trimmed = WorldFeatLabelGen.trim_to_shortest(list(feats))
    out["tts_0"], out["tts_1"], out["tts_3"] = trimmed[0], trimmed[1], trimmed[3]
    # sequence_mask both layouts
    lens = torch.tensor([5, 2, 7, 1])
    out["sm_len"] = lens.numpy()
    out["sm_bf"] = Handler.sequence_mask(lens, 7, batch_first=True).numpy()
    out["sm_tf"] = Handler.sequence_mask(lens, 7, batch_first=False).numpy()
    # NamedLoss mean_per_frame (time-major, as the trainers feed it by default)
    from idiaptts.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    cfg = NamedLoss.Config(name="MSELoss_acoustic_features", type_="MSELoss",
                           seq_mask="acoustic_features_mask",
                           input_names=["acoustic_features", "pred_acoustic_features"],
                           batch_first=False)
    loss_mod = cfg.create_loss()
    T, B, D = 9, 3, 6
    lens = torch.tensor([9, 4, 6])
    tgt = torch.from_numpy(rng.normal(size=(T, B, D)).astype(np.float32))
    pred = torch.from_numpy(rng.normal(size=(T, B, D)).astype(np.float32)).requires_grad_(True)
    mask = Handler.sequence_mask(lens, T, batch_first=False)
    data = {"acoustic_features": tgt, "pred_acoustic_features": pred,
            "acoustic_features_mask": mask}
    ld = loss_mod(data, {"acoustic_features_mask": lens}, step=1)
    val = list(ld.values())[0]
    val.backward()
    out["nl_target"], out["nl_pred"], out["nl_len"] = tgt.numpy(), pred.detach().numpy(), lens.numpy()
    out["nl_loss"], out["nl_grad"] = val.detach().numpy(), pred.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "host_logic.npz"), **out)
    print("host_logic.npz:", len(out), "arrays")


def capture_benchmark_kat():
    """Reference benchmark known answer with oracle MLPG as the bandmat stand-in."""
    import idiaptts.misc.mlpg as ref_mlpg
    import idiaptts.src.Metrics as ref_metrics
    from idiaptts.src.model_trainers.AcousticModelTrainer import AcousticModelTrainer
    from oracle import capi

    calls = []

    def generation(self, features, covariance, feature_dim):
        var = np.ascontiguousarray(np.diag(covariance), dtype=np.float64)
        res = capi.mlpg(np.asarray(features, dtype=np.float64), var, feature_dim)
        calls.append((np.array(features), var, feature_dim, res))
        return res
    ref_mlpg.MLPG.generation = generation
    ref_metrics.nnmnkwii_metrics.melcd = lambda X, Y, lengths=None: float(
        10.0 / np.log(10) * np.sqrt(2.0) * np.sqrt(((X - Y) ** 2).sum(-1)).mean())

    os.chdir(os.path.join(REF, "test"))
    hp = AcousticModelTrainer.create_hparams()
    hp.num_questions = 409
    hp.voice = "full"
    hp.data_dir = os.path.realpath(os.path.join("integration", "fixtures", "database"))
    hp.out_dir = "/tmp/idiaptts_amd_golden_benchmark"
    hp.frame_size_ms = 5
    hp.num_coded_sps = 20
    hp.seed = 1
    hp.epochs = 3
    hp.use_gpu = False
    hp.model_type = "RNNDYN-1_RELU_32-1_FC_67"
    hp.batch_size_train = 2
    hp.batch_size_val = 50
    hp.use_saved_learning_rate = True
    hp.optimiser_args["lr"] = 0.001
    hp.model_name = "test_model"
    hp.epochs_per_checkpoint = 2
    hp.world_dir = os.path.join("integration", "fixtures", "WORLD")
    with open(os.path.join("integration", "fixtures", "database", "file_id_list.txt")) as f:
        id_list = [s.strip() for s in f.readlines()]
    trainer = AcousticModelTrainer(**AcousticModelTrainer.legacy_support_init(
        hp.world_dir, os.path.join("integration", "fixtures", "questions"), id_list,
        hp.num_questions, hp))
    trainer.init(hp)
    # end-to-end capture: what the test utterance looks like at every stage boundary
    e2e = {}
    wreader = trainer.datareaders["acoustic_features"]
    qreader = trainer.datareaders["questions"]
    from idiaptts.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen as RefW
    orig_post = RefW._postprocess_world

    def rec_post(self, sample, *a, **k):
        e2e["denormalised"] = np.array(sample)
        return orig_post(self, sample, *a, **k)
    RefW._postprocess_world = rec_post
    orig_inf = trainer.model_handler.inference

    def rec_inf(data, *a, **k):
        e2e["questions_norm"] = np.array(data["questions"].detach().cpu().numpy()
                                         if hasattr(data["questions"], "detach")
                                         else data["questions"])
        out = orig_inf(data, *a, **k)
        return out
    trainer.model_handler.inference = rec_inf
    scores = trainer.benchmark(hp)["pred_acoustic_features"]
    RefW._postprocess_world = orig_post
    e2e["test_id"] = np.array(trainer.id_list_test[0])
    e2e["out_mean"] = np.asarray(wreader.norm_params[0], dtype=np.float64)
    e2e["out_std"] = np.asarray(wreader.norm_params[1], dtype=np.float64)
    for i, c in enumerate(wreader.covs):
        if c is not None:
            e2e["cov_%d" % i] = np.asarray(c, dtype=np.float64)
    for k, v in trainer.model_handler.model.state_dict().items():
        e2e["sd_" + k] = v.cpu().numpy()
    e2e["original"] = np.asarray(trainer.get_output_dict(trainer.id_list_test, hp)[trainer.id_list_test[0]])
    e2e["scores"] = np.array(scores, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "benchmark_e2e.npz"), **e2e)
    print("benchmark_e2e.npz:", {k: getattr(v, "shape", None) for k, v in e2e.items() if not k.startswith("sd_")})
    shutil.rmtree(hp.out_dir, ignore_errors=True)
    print("benchmark scores", scores)
    np.testing.assert_almost_equal((8.616, 78.4, 0.609, 37.352), scores, 3)
    out = {"scores": np.array(scores, dtype=np.float64), "n_calls": np.array(len(calls))}
    for i, (f, v, d, r) in enumerate(calls):
        out["feat_%d" % i] = f.astype(np.float32) if f.dtype == np.float32 else f
        out["var_%d" % i] = v
        out["dim_%d" % i] = np.array(d)
        out["out_%d" % i] = r
    np.savez_compressed(os.path.join(HERE, "mlpg_benchmark_kat.npz"), **out)
    print("mlpg_benchmark_kat.npz:", len(calls), "MLPG calls")


def _ref_hparams(AcousticModelTrainer, out_dir):
    hp = AcousticModelTrainer.create_hparams()
    hp.num_questions = 409
    hp.voice = "full"
    hp.data_dir = os.path.realpath(os.path.join("integration", "fixtures", "database"))
    hp.out_dir = out_dir
    hp.frame_size_ms = 5
    hp.num_coded_sps = 20
    hp.seed = 1
    hp.epochs = 3
    hp.use_gpu = False
    hp.model_type = "RNNDYN-1_RELU_32-1_FC_67"
    hp.batch_size_train = 2
    hp.batch_size_val = 50
    hp.use_saved_learning_rate = True
    hp.optimiser_args["lr"] = 0.001
    hp.model_name = "test_model"
    hp.epochs_per_checkpoint = 2
    hp.world_dir = os.path.join("integration", "fixtures", "WORLD")
    return hp


def capture_trainer_fixture():
    """Data the reference's trainer tests read (test/integration/fixtures/{questions,WORLD/
    cmp_mcep20,database/file_id_list.txt}: 9 utterances of raw-f32 `.questions` / `.cmp` plus the
    legacy `.bin` normalisation files) packed into one compressed archive, and the losses of the
    reference's own test_train run (test_AcousticModelTrainer.py:75-92: seed 1234, 3 epochs,
    CPU) for the trainer parity test."""
    from idiaptts.src.model_trainers.AcousticModelTrainer import AcousticModelTrainer
    os.chdir(os.path.join(REF, "test"))
    with open(os.path.join("integration", "fixtures", "database", "file_id_list.txt")) as f:
        id_list = [s.strip() for s in f.readlines()]
    out = {"id_list": np.array(id_list)}
    for i in id_list:
        q = np.fromfile(os.path.join(FIX, "questions", i + ".questions"), dtype=np.float32)
        out["questions/" + i] = q.reshape(-1, 409)
        out["cmp/" + i] = np.fromfile(os.path.join(FIX, "WORLD", "cmp_mcep20", i + ".cmp"),
                                      dtype=np.float32).reshape(-1, 67)
    out["bin/questions/min-max.bin"] = np.fromfile(os.path.join(FIX, "questions", "min-max.bin"),
                                                   dtype=np.uint8)
    for n in ["mcep20", "lf0", "bap"]:
        name = n + "-mean-covariance.bin"
        out["bin/WORLD/cmp_mcep20/" + name] = np.fromfile(
            os.path.join(FIX, "WORLD", "cmp_mcep20", name), dtype=np.uint8)

    out_dir = "/tmp/idiaptts_amd_golden_train"
    for seed, tag in [(1234, "train"), (1, "seed1")]:
        hp = _ref_hparams(AcousticModelTrainer, out_dir)
        hp.seed = seed
        hp.use_best_as_final_model = False
        trainer = AcousticModelTrainer(**AcousticModelTrainer.legacy_support_init(
            hp.world_dir, os.path.join("integration", "fixtures", "questions"), id_list,
            hp.num_questions, hp))
        trainer.init(hp)
        out[tag + "_ids_train"] = np.array(trainer.id_list_train)
        out[tag + "_ids_val"] = np.array(trainer.id_list_val)
        out[tag + "_ids_test"] = np.array(trainer.id_list_test)
        for k, v in trainer.model_handler.model.state_dict().items():
            out[tag + "_init/" + k] = np.array(v.cpu().numpy(), copy=True)
        if tag == "train":
            all_loss, all_loss_train, _ = trainer.train(hp)
            key = "MSELoss_acoustic_features"
            out["train_val_losses"] = np.asarray(all_loss[key], dtype=np.float64)
            out["train_train_losses"] = np.asarray(all_loss_train[key], dtype=np.float64)
            for k, v in trainer.model_handler.model.state_dict().items():
                out["train_final/" + k] = v.cpu().numpy()
            print("reference losses: val", out["train_val_losses"], "train",
                  out["train_train_losses"])
        shutil.rmtree(out_dir, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, "trainer_fixture.npz"), **out)
    print("trainer_fixture.npz:", os.path.getsize(os.path.join(HERE, "trainer_fixture.npz")), "bytes")


def capture_duration_fixture():
    """Data the reference's (commented-out) duration-trainer tests read -- fixtures/dur/*.dur with
    mean-std_dev.bin, labels/mono_no_align/*.lab, labels/mono_phone.list -- and what the
    reference's PhonemeLabelGen returns for both label layouts."""
    from idiaptts.src.data_preparation.phonemes.PhonemeLabelGen import PhonemeLabelGen
    import idiaptts.src.Metrics as ref_metrics
    import scipy.stats
    with open(os.path.join(FIX, "database", "file_id_list.txt")) as f:
        id_list = [s.strip() for s in f.readlines()]
    out = {"id_list": np.array(id_list)}
    with open(os.path.join(FIX, "labels", "mono_phone.list")) as f:
        out["mono_phone_list"] = np.array(f.read())
    out["bin/dur/mean-std_dev.bin"] = np.fromfile(os.path.join(FIX, "dur", "mean-std_dev.bin"), dtype=np.uint8)
    symbol_dict = PhonemeLabelGen.get_symbol_dict(os.path.join(FIX, "labels", "mono_phone.list"))
    out["symbols"] = np.array(list(symbol_dict.keys()))
    out["symbol_ids"] = np.array(list(symbol_dict.values()))
    for i in id_list:
        out["dur/" + i] = np.fromfile(os.path.join(FIX, "dur", i + ".dur"), dtype=np.float32).reshape(-1, 5)
        with open(os.path.join(FIX, "labels", "mono_no_align", i + ".lab")) as f:
            out["mono_no_align/" + i] = np.array(f.read())
        for ltype, sub in (("full_state_align", "label_state_align"), ("mono_no_align", "mono_no_align")):
            ids = PhonemeLabelGen.load_sample(i, os.path.join(FIX, "labels", sub), symbol_dict, ltype)
            out["ids_%s/%s" % (ltype, i)] = np.asarray(ids)
    rng = np.random.default_rng(3)
    a, b = rng.normal(size=(40, 5)), rng.normal(size=(40, 5))
    ref_metrics.scipy = sys.modules.get("scipy", __import__("scipy"))
    out["metric_a"], out["metric_b"] = a, b
    out["metric_rmse"] = ref_metrics.Metrics.rmse(a, b)
    out["metric_pearson"] = np.array([scipy.stats.pearsonr(a[:, k], b[:, k])[0] for k in range(5)])
    np.savez_compressed(os.path.join(HERE, "duration_fixture.npz"), **out)
    print("duration_fixture.npz:", os.path.getsize(os.path.join(HERE, "duration_fixture.npz")), "bytes;",
          len(symbol_dict), "symbols")


def capture_embedding_model():
    """reference RNNDyn with an embedding group (rnn_dyn/RNNDyn.py:39-49, 88-134): index in the
    last input column, embedding concatenated in front of groups 0 and 2 -> model_embedding.npz"""
    import torch
    from idiaptts.src.neural_networks.pytorch.models.rnn_dyn.Config import Config
    torch.manual_seed(9)
    cfg = Config(in_dim=6, batch_first=False, layer_configs=[
        Config.LayerConfig(layer_type="Linear", out_dim=8, num_layers=1, nonlin="tanh"),
        Config.LayerConfig(layer_type="GRU", out_dim=16, num_layers=1, bidirectional=True),
        Config.LayerConfig(layer_type="Linear", out_dim=4, num_layers=1)],
        emb_configs=[Config.EmbeddingConfig(embedding_dim=3, name="emb_speaker", num_embedding=4,
                                            affected_layer_group_indices=(0, 2))])
    model = cfg.create_model()
    lens = torch.tensor([7, 3, 5])
    x = torch.randn(7, 3, 6)
    idx = torch.tensor([2.0, 0.0, 3.0]).view(1, 3, 1).expand(7, 3, 1)
    w = torch.randn(7, 3, 4)
    for b, l in enumerate(lens):
        w[l:, b] = 0
    model.init_hidden(3)
    out, _ = model(torch.cat((x, idx), dim=2), seq_lengths_input=lens, max_length_inputs=7)
    (out * w).sum().backward()
    res = {"x": x.numpy(), "idx": idx.contiguous().numpy(), "w": w.numpy(), "len": lens.numpy(),
           "out": out.detach().numpy()}
    for k, v in model.state_dict().items():
        res["sd_" + k] = v.numpy()
    for k, p_ in model.named_parameters():
        res["grad_" + k] = p_.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "model_embedding.npz"), **res)
    print("model_embedding.npz:", sorted(res))


def npz_reader_cases(tmp):
    """Writes the seeded archives + parameter files of the NpzDataReader cases below `tmp` and
    returns {case: Config kwargs}; shared by the capture (reference reader) and the test (ours)."""
    rng = np.random.default_rng(42)
    d1, d2 = os.path.join(tmp, "a"), os.path.join(tmp, "b")
    os.makedirs(d1, exist_ok=True)
    os.makedirs(d2, exist_ok=True)
    for k, n in enumerate(["utt1", "utt2"]):
        T = 17 + 6 * k
        np.savez(os.path.join(d1, n), cmp=rng.normal(size=(T, 7)), dur=rng.integers(1, 9, (T, 2)))
        np.savez(os.path.join(d2, n), lf0=rng.normal(size=(T, 2)).astype(np.float32))
    mean, std = rng.normal(size=7), rng.uniform(0.5, 2.0, size=7)
    np.savez(os.path.join(d1, "set-mean-std_dev"), mean=mean, std_dev=std, sum_length=np.array(40))
    mn, mx = -rng.uniform(1, 2, size=2), rng.uniform(1, 2, size=2)
    np.savez(os.path.join(d2, "min-max"), min=mn, max=mx)
    return {
        "plain": dict(name="cmp", directory=d1),
        "stddev_file": dict(name="cmp", directory=d1, norm_type="MEAN_STDDEV",
                            norm_file=(d1, "set")),
        "indices": dict(name="cmp", directory=d1, indices=np.array([5, 0, 3]),
                        norm_type="MEAN_STDDEV",
                        norm_params=(mean[[5, 0, 3]], std[[5, 0, 3]])),
        "two_dirs": dict(name="both", directory=[d1, d2], features=["lf0", "dur"],
                         output_names=["f", "d"]),
        "fn_before": dict(name="cmp", directory=d1, norm_type="MEAN_STDDEV",
                          norm_params=(mean, std), preprocessing_fn="square",
                          preprocess_before_norm=True, postprocessing_fn="square",
                          postprocess_before_norm=False),
        "minmax_chunk": dict(name="lf0", directory=d2, norm_type="MIN_MAX", norm_file=(None, None),
                             chunk_size=4, pad_mode="edge"),
    }


def run_npz_reader_cases(NpzDataReader, tmp):
    """{case/id/output: array} for both ids through reader[id] and postprocess_sample."""
    out = {}
    for case, kw in npz_reader_cases(tmp).items():
        kw = dict(kw)
        norm_file = kw.pop("norm_file", None)
        kw["norm_type"] = getattr(NpzDataReader.Config.NormType, kw.pop("norm_type", "NONE"))
        for fn in ("preprocessing_fn", "postprocessing_fn"):
            if kw.get(fn) == "square":
                kw[fn] = np.square
        reader = NpzDataReader.Config(**kw).create_reader()
        if norm_file is not None:
            reader.get_normalisation_params(*norm_file)
        for n in ["utt1", "utt2"]:
            item = reader[n]
            for name in reader.output_names:
                out["{}/{}/{}".format(case, n, name)] = item[name]
            out["{}/{}/len".format(case, n)] = np.array(reader.get_length(n))
            if len(reader.output_names) == 1:
                out["{}/{}/post".format(case, n)] = np.asarray(
                    reader.postprocess_sample(item[reader.output_names[0]]))
    return out


def capture_npz_reader():
    """reference NpzDataReader (data_preparation/NpzDataReader.py) on seeded archives ->
    npz_reader_fixture.npz"""
    import tempfile
    np.long = np.int64        # removed from numpy; the reference's Config casts indices with it
    from idiaptts.src.data_preparation.NpzDataReader import NpzDataReader
    with tempfile.TemporaryDirectory() as tmp:
        out = run_npz_reader_cases(NpzDataReader, tmp)
    np.savez_compressed(os.path.join(HERE, "npz_reader_fixture.npz"), **out)
    print("wrote npz_reader_fixture.npz:", len(out), "arrays")


def _main():
    copy_data_fixtures()
    install_stub_harness()
    if "--models" in sys.argv:
        capture_reference_model_forward()
        return
    if "--kat" in sys.argv:
        capture_benchmark_kat()
        return
    if "--trainer" in sys.argv:
        capture_trainer_fixture()
        return
    if "--duration" in sys.argv:
        capture_duration_fixture()
        return
    if "--npz-reader" in sys.argv:
        capture_npz_reader()
        return
    if "--embedding" in sys.argv:
        capture_embedding_model()
        return
    if "--config3" in sys.argv:
        capture_config3_step()
        return
    capture_host_logic()
    capture_benchmark_kat()
    capture_reference_model_forward()


def capture_reference_model_forward():
    """Forward / loss / gradients of the reference's own RNNDyn + NamedForwardWrapper + NamedLoss
    (torch CPU) for (a) the fixture checkpoint test_model_in409_out67 on fixture questions and
    (b) a small seeded BiLSTM model on a padded batch -- inputs, state dict and outputs stored."""
    import torch
    from idiaptts.src.neural_networks.pytorch.models import rnn_dyn
    from idiaptts.src.neural_networks.pytorch.models.NamedForwardWrapper import NamedForwardWrapper
    from idiaptts.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    from idiaptts.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler
    out = {}
    # (a) fixture checkpoint: RNNDYN-1_RELU_32-1_FC_67, 409 -> 67
    hp = types.SimpleNamespace(model_type="RNNDYN-1_RELU_32-1_FC_67", batch_first=False,
                               dropout=0.0)
    cfg = NamedForwardWrapper.Config(rnn_dyn.convert_legacy_to_config((409,), hp),
                                     input_names=["questions"], batch_first=False,
                                     name="AcousticModel", output_names=["pred_acoustic_features"])
    model = cfg.create_model()
    ck = torch.load(os.path.join(FIX, "test_model_in409_out67", "nn", "params_best"),
                    weights_only=False)
    model.load_state_dict(ck["params"])
    q = np.fromfile(os.path.join(FIX, "questions", "LJ001-0008.questions"),
                    dtype=np.float32).reshape(-1, 409)[:120]
    data = {"questions": torch.from_numpy(q[:, None, :].copy())}
    lengths = {"questions": torch.tensor([120])}
    model.init_hidden(1)
    model(data, lengths, {"questions": 120})
    out["a_questions"] = q
    out["a_pred"] = data["pred_acoustic_features"].detach().numpy()
    for k, v in ck["params"].items():
        out["a_sd_" + k] = v.numpy()
    # (b) seeded small BiLSTM: 2_TANH_24-2_BiLSTM_16-1_FC_7 on a ragged batch, time-major
    torch.manual_seed(5)
    hp = types.SimpleNamespace(model_type="RNNDYN-2_TANH_24-2_BiLSTM_16-1_FC_7", batch_first=False,
                               dropout=0.0)
    cfg = NamedForwardWrapper.Config(rnn_dyn.convert_legacy_to_config((11,), hp),
                                     input_names=["questions"], batch_first=False, name="AM",
                                     output_names=["pred_acoustic_features"])
    model = cfg.create_model()
    lens = torch.tensor([9, 4, 7])
    x = torch.randn(9, 3, 11)
    tgt = torch.randn(9, 3, 7)
    for b, l in enumerate(lens):
        x[l:, b] = 0
        tgt[l:, b] = 0
    mask = Handler.sequence_mask(lens, 9, batch_first=False)
    data = {"questions": x, "acoustic_features": tgt, "acoustic_features_mask": mask}
    lengths = {"questions": lens, "acoustic_features": lens, "acoustic_features_mask": lens}
    model.init_hidden(3)
    model(data, lengths, {"questions": 9})
    loss_mod = NamedLoss.Config(name="MSELoss_acoustic_features", type_="MSELoss",
                                seq_mask="acoustic_features_mask",
                                input_names=["acoustic_features", "pred_acoustic_features"],
                                batch_first=False).create_loss()
    loss = list(loss_mod(data, lengths, step=1).values())[0]
    loss.backward()
    out["b_x"], out["b_tgt"], out["b_len"] = x.numpy(), tgt.numpy(), lens.numpy()
    out["b_pred"] = data["pred_acoustic_features"].detach().numpy()
    out["b_loss"] = loss.detach().numpy()
    for k, v in model.state_dict().items():
        out["b_sd_" + k] = v.numpy()
    for k, p in model.named_parameters():
        out["b_grad_" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "model_forward.npz"), **out)
    print("model_forward.npz:", len(out), "arrays")


def capture_config3_step():
    """One training step of the reference's own module stack at the BASELINE config-3 size
    (RNNDYN-3_BiLSTM_512-1_FC_187, 425 -> 187, B = 33 ragged rows, torch CPU fp32): forward,
    NamedLoss(MSELoss x mask, mean per frame), backward, torch.optim.Adam(lr 1e-3) step -- the
    calls of ModularModelHandlerPyTorch.py:745-831.  Weights / batch are regenerated from seeds
    (tests/config3_data.py); stored: prediction of two rows, loss, per-parameter gradient norm,
    sum and sampled entries, sampled parameters after the step."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import config3_data as c3
    from idiaptts.src.neural_networks.pytorch.models import rnn_dyn
    from idiaptts.src.neural_networks.pytorch.models.NamedForwardWrapper import NamedForwardWrapper
    from idiaptts.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    from idiaptts.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler
    torch.set_num_threads(8)
    hp = types.SimpleNamespace(model_type=c3.MODEL_TYPE, batch_first=False, dropout=0.0)
    cfg = NamedForwardWrapper.Config(rnn_dyn.convert_legacy_to_config((c3.IN_DIM,), hp),
                                     input_names=["questions"], batch_first=False, name="AM",
                                     output_names=["pred_acoustic_features"])
    model = cfg.create_model()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c3.state(shapes).items()})
    x, y, lens = c3.batch()
    lens_t = torch.from_numpy(lens)
    T, B = x.shape[:2]
    data = {"questions": torch.from_numpy(x), "acoustic_features": torch.from_numpy(y),
            "acoustic_features_mask": Handler.sequence_mask(lens_t, T, batch_first=False)}
    lengths = {"questions": lens_t, "acoustic_features": lens_t, "acoustic_features_mask": lens_t}
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    model.init_hidden(B)
    model(data, lengths, {"questions": T})
    loss_mod = NamedLoss.Config(name="MSELoss_acoustic_features", type_="MSELoss",
                                seq_mask="acoustic_features_mask",
                                input_names=["acoustic_features", "pred_acoustic_features"],
                                batch_first=False).create_loss()
    loss = list(loss_mod(data, lengths, step=1).values())[0]
    loss.backward()
    out = {"loss": loss.detach().numpy(), "lens": lens,
           "pred_rows": data["pred_acoustic_features"].detach().numpy()[:, [0, B - 1]],
           "keys": np.array(sorted(shapes))}
    for k, p in model.named_parameters():
        g = p.grad.numpy().reshape(-1)
        idx = c3.sample_index(k, g.size)
        out["gnorm_" + k] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out["gsum_" + k] = np.array(g.astype(np.float64).sum())
        out["gsamp_" + k] = g[idx]
    opt.step()
    for k, p in model.named_parameters():
        out["psamp_" + k] = p.detach().numpy().reshape(-1)[c3.sample_index(k, p.numel())]
    np.savez_compressed(os.path.join(HERE, "config3_step.npz"), **out)
    print("config3_step.npz: loss", float(loss), "params", sum(int(np.prod(s)) for s in shapes.values()))


if __name__ == "__main__":
    _main()
