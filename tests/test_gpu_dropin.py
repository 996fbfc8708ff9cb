"""Drop-in surface on the GPU: the reference's own integration checks for the WORLD path
(test/integration/data_preparation/world/test_WorldFeatLabelGen.py) restated against our shims."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_world_extract_and_resynth_bounds(gpu, golden_dir):
    """reference test_WorldFeatLabelGen.py:722-763: shapes/dtypes of world_extract_features and
    sum (orig - WORLD resynth)^2 < 10000."""
    from idiaptts_amd.src.data_preparation.audio.AudioProcessing import AudioProcessing
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    raw, fs = AudioProcessing.get_raw(os.path.join(golden_dir, "LJ001-0008.wav"), 0.0)
    amp_sp, lf0, vuv, bap = WorldFeatLabelGen.world_extract_features(raw, fs, 5)
    T = int(1000.0 * len(raw) / fs / 5) + 1
    assert amp_sp.shape == (T, 513) and amp_sp.dtype == np.float64
    assert lf0.shape == (T, 1) and lf0.dtype == np.float32
    assert vuv.shape == (T, 1) and vuv.dtype == np.float32 and set(np.unique(vuv)) <= {0.0, 1.0}
    assert bap.shape == (T, 1) and bap.dtype == np.float32
    wav = WorldFeatLabelGen.world_features_to_raw(amp_sp, lf0, vuv, bap, fs=fs, n_fft=1024)
    n = min(len(wav), len(raw))
    assert ((raw[:n] - wav[:n]) ** 2).sum() < 10000
    # mcep80 -> spectrum reconstruction bound (reference :816-824): sum err^2 < 100
    mcep = AudioProcessing.extract_mcep(amp_sp, 80, 0.42)
    assert mcep.dtype == np.float32 and mcep.shape == (T, 80)
    rec = AudioProcessing.mcep_to_amp_sp(mcep, fs, 0.42)
    assert rec.dtype == np.float32 and ((amp_sp - rec) ** 2).sum() < 100


def test_gen_data_matches_reference_cmp_and_roundtrips(gpu, golden_dir, tmp_path):
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    ids = ["LJ001-000%d" % i for i in range(1, 10)]   # every fixture utterance of the reference
    gen = WorldFeatLabelGen(str(tmp_path), add_deltas=True, preemphasis=0.97, num_coded_sps=20,
                            mgc_alpha=0.58)
    label_dict, mean, cov = gen.gen_data(golden_dir, str(tmp_path), "ids.txt", id_list=ids,
                                         return_dict=True)
    # normalisation statistics of the nine utterances (itts_feature_stats: sum x, sum x x^T in fp64 on
    # the device) against the reference-held WORLD/cmp_mcep20/<feat>-mean-covariance.bin (made with
    # float32 sums: agreement to their rounding)
    import struct
    for k, feat in ((0, "mcep20"), (1, "lf0"), (3, "bap")):
        with open(os.path.join(golden_dir, "stats", feat + "-mean-covariance.bin"), "rb") as f:
            n_frames, size = struct.unpack("ii", f.read(8))
            ref = np.fromfile(f, dtype=np.float64).reshape(size, -1)
        assert n_frames == sum(len(v) for v in label_dict.values())
        assert np.abs(np.asarray(mean[k]).ravel() - ref[0]).max() < 5e-6
        assert np.abs(np.asarray(cov[k]) - ref[1:]).max() < 5e-5 * max(1.0, np.abs(ref[1:]).max())
    for n in ids:
        cmp_ = np.fromfile(os.path.join(golden_dir, n + ".cmp"), dtype=np.float32).reshape(-1, 67)
        got = label_dict[n]
        assert got.shape == cmp_.shape
        assert np.array_equal(got[:, 63], cmp_[:, 63])                  # V/UV bit-exact
        assert np.sqrt(np.mean((got - cmp_) ** 2)) < 1e-6               # bar: 1e-4 RMSE
        assert np.array_equal(gen.load(n), got)                          # npz round trip
    assert os.path.isfile(os.path.join(str(tmp_path), "mcep20", "ids-deltas-mean-covariance.npz"))
    assert os.path.isfile(os.path.join(str(tmp_path), "lf0", "ids-deltas-stats.npz"))
    assert len(mean) == 4 and cov[0].shape == (60, 60)
    # legacy cmp fallback of load(): directory layout <dir>/cmp_mcep20/<id>.cmp
    os.makedirs(os.path.join(str(tmp_path), "legacy", "cmp_mcep20"))
    import shutil
    shutil.copy(os.path.join(golden_dir, "LJ001-0008.cmp"),
                os.path.join(str(tmp_path), "legacy", "cmp_mcep20", "LJ001-0008.cmp"))
    leg = WorldFeatLabelGen(os.path.join(str(tmp_path), "legacy"), add_deltas=True,
                            num_coded_sps=20)
    cmp_ = np.fromfile(os.path.join(golden_dir, "LJ001-0008.cmp"), dtype=np.float32).reshape(-1, 67)
    assert np.array_equal(leg.load("LJ001-0008"), cmp_)


def test_postprocess_world_mlpg_and_run_world_synth(gpu, golden_dir, tmp_path):
    from idiaptts_amd.src.Synthesiser import Synthesiser
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    from oracle import capi
    cmp_ = np.fromfile(os.path.join(golden_dir, "LJ001-0008.cmp"), dtype=np.float32).reshape(-1, 67)
    gen = WorldFeatLabelGen(None, add_deltas=True, num_coded_sps=20)
    rng = np.random.default_rng(0)
    gen.covs = [np.diag(rng.uniform(0.05, 1, 60)), np.diag(rng.uniform(0.05, 1, 3)), None,
                np.diag(rng.uniform(0.05, 1, 3))]
    sample = cmp_.astype(np.float64).copy()
    out = gen._postprocess_world(sample)
    assert out.shape == (cmp_.shape[0], 23)
    ref_sp = capi.mlpg(cmp_[:, :60].astype(np.float64), np.diag(gen.covs[0]), 20)
    ref_lf0 = capi.mlpg(cmp_[:, 60:63].astype(np.float64), np.diag(gen.covs[1]), 1)
    ref_bap = capi.mlpg(cmp_[:, 64:67].astype(np.float64), np.diag(gen.covs[3]), 1)
    assert np.abs(out[:, :20] - ref_sp).max() < 1e-9
    assert np.abs(out[:, 20:21] - ref_lf0).max() < 1e-9
    assert np.array_equal(out[:, 21], cmp_[:, 63])
    assert np.abs(out[:, 22:23] - ref_bap).max() < 1e-9
    # run_world_synth on static features: wav written with the reference's file name and length
    hp = types.SimpleNamespace(synth_fs=16000, num_coded_sps=20, num_bap=1, sp_type="mcep",
                               out_dir=str(tmp_path), model_name="m", synth_file_suffix="_x",
                               synth_ext="wav", do_post_filtering=False)
    wavs = Synthesiser.run_world_synth({"LJ001-0008": out.astype(np.float32)}, hp,
                                       return_waveforms=True)
    path = os.path.join(str(tmp_path), "m", "synth", "LJ001-0008_x_20mcep_WORLD.wav")
    assert os.path.isfile(path)
    # length check of the reference's synth test (test_AcousticModelTrainer.py:161-168)
    assert len(wavs["LJ001-0008"]) == int(cmp_.shape[0] * 5 * 16000 / 1000)
    assert Synthesiser.synth_world_features is not None


def test_lf0_label_gen_matches_oracle_dio_stonemask(gpu, golden_dir, tmp_path):
    """LF0LabelGen.gen_data (reference LF0LabelGen.py:209-322: pyworld.dio + stonemask on the raw
    wav, log-F0, 20 Hz threshold, interpolate_lin) against the C oracle's DIO + StoneMask: identical
    V/UV, lf0 within 1e-6; the legacy files and normalisation parameters read back."""
    import shutil
    from oracle import capi
    from idiaptts_amd.misc.utils import interpolate_lin
    from idiaptts_amd.src.data_preparation.audio.AudioProcessing import AudioProcessing
    from idiaptts_amd.src.data_preparation.world.LF0LabelGen import LF0LabelGen
    wav_dir, out_dir = str(tmp_path / "wav"), str(tmp_path / "out")
    os.makedirs(wav_dir)
    ids = ["LJ001-0002", "LJ001-0008"]
    for i in ids:
        shutil.copy(os.path.join(golden_dir, i + ".wav"), wav_dir)
    gen = LF0LabelGen(out_dir)
    labels, mean, std = gen.gen_data(wav_dir, out_dir, "ids.txt", ids, return_dict=True)
    assert mean.shape == (2,) and mean[1] == 0.0 and std[1] == 1.0
    for i in ids:
        raw, fs = AudioProcessing.get_raw(os.path.join(wav_dir, i + ".wav"))
        f0, tp = capi.dio(raw, fs)
        f0 = capi.stonemask(raw, fs, tp, f0)
        with np.errstate(divide="ignore"):
            lf0 = np.log(f0).astype(np.float32)
        lf0[lf0 <= np.log(20)] = 0
        ref_lf0, ref_vuv = interpolate_lin(lf0)
        got = labels[i]
        assert got.shape == (len(f0), 2) and got.dtype == np.float32
        assert np.array_equal(got[:, 1:], ref_vuv.astype(np.float32))
        assert np.abs(got[:, :1] - ref_lf0).max() < 1e-6
        assert np.array_equal(LF0LabelGen.load_sample(i, out_dir), got)
    gen.get_normalisation_params(out_dir, "ids")
    x = gen[ids[0]]
    assert x.dtype == np.float32 and np.allclose(gen.postprocess_sample(x), labels[ids[0]], atol=1e-5)
    d, dm, ds = LF0LabelGen(out_dir, add_deltas=True).gen_data(wav_dir, out_dir, "ids.txt", ids,
                                                               add_deltas=True, return_dict=True)
    assert d[ids[1]].shape[1] == 4 and dm[-1] == 0.0 and abs(ds[-1] - 1.0) < 1e-12
    assert np.array_equal(LF0LabelGen.load_sample(ids[1], out_dir, add_deltas=True), d[ids[1]])


def _gen_data_worker(rank, world, port, wav_dir, out_dir, ids, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    gen = WorldFeatLabelGen(out_dir, add_deltas=True, num_coded_sps=20)
    label_dict, mean, cov = gen.gen_data(wav_dir, out_dir, "ids.txt", id_list=ids, return_dict=True)
    ret[rank] = (list(label_dict.keys()), [np.asarray(m) for m in mean], [np.asarray(c) for c in cov])
    dist.destroy_process_group()


def test_gen_data_sharded_over_two_ranks_equals_single_process(gpu, golden_dir, tmp_path):
    """utterances are partitioned over the ranks (one process per GPU; here
    two processes share the one GPU of the test box and talk over gloo), each analyses and writes
    its own, the normalisation statistics are merged by a sum all-reduce.  Files and parameters
    must equal the single-process run."""
    import shutil
    import socket
    import torch.multiprocessing as mp
    from scipy.io import wavfile
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    ids = ["LJ001-0002", "LJ001-0008"]
    for n in ids:
        shutil.copy(os.path.join(golden_dir, n + ".wav"), str(wav_dir / (n + ".wav")))
    fs, w = wavfile.read(os.path.join(golden_dir, "LJ001-0008.wav"))
    for k, (a, b) in enumerate([(0, 30000), (9000, 21000), (15000, 60000)]):   # ragged extra clips
        wavfile.write(str(wav_dir / "clip{}.wav".format(k)), fs, w[a:b])
        ids.append("clip{}".format(k))
    single = tmp_path / "single"
    gen = WorldFeatLabelGen(str(single), add_deltas=True, num_coded_sps=20)
    ref_dict, ref_mean, ref_cov = gen.gen_data(str(wav_dir), str(single), "ids.txt", id_list=ids,
                                               return_dict=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    sharded = tmp_path / "sharded"
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_gen_data_worker, args=(2, port, str(wav_dir), str(sharded), ids, ret), nprocs=2,
             join=True)
    for rank in (0, 1):
        keys, mean, cov = ret[rank]
        assert keys == ids                                               # same order on every rank
        # (like the reference, the extractors sum float32 features per utterance: the grouping of
        # the partial sums shows at float32 resolution)
        for m, r in zip(mean, ref_mean):
            assert np.allclose(m, r, rtol=1e-5, atol=1e-6)
        for c, r in zip(cov, ref_cov):
            assert np.allclose(c, r, rtol=1e-4, atol=1e-5)
    reader = WorldFeatLabelGen(str(sharded), add_deltas=True, num_coded_sps=20)
    for n in ids:
        assert np.array_equal(reader.load(n), ref_dict[n])               # feature files identical
    assert os.path.isfile(str(sharded / "mcep20" / "ids-deltas-mean-covariance.npz"))


def test_merlin_post_filter_and_decode_sp(gpu, golden_dir):
    """decode_sp(post_filtering=True) (AudioProcessing.py:303-314): the post filter against the
    numpy restatement (nnmnkwii is not vendored: parity unpinned), energy preserved, formants
    sharpened; then the decoded spectrum."""
    from idiaptts_amd.src.data_preparation.audio.AudioProcessing import (AudioProcessing,
                                                                         merlin_post_filter)
    from oracle import world_spec as ws
    cmp_ = np.fromfile(os.path.join(golden_dir, "LJ001-0008.cmp"), dtype=np.float32).reshape(-1, 67)
    mc = cmp_[100:140, :20].astype(np.float64)
    alpha = 0.41
    got = merlin_post_filter(mc, alpha)
    want = ws.merlin_post_filter(mc, alpha)
    assert np.abs(got - want).max() < 1e-10
    sp0 = AudioProcessing.decode_sp(mc, "mcep", 16000, alpha)
    sp1 = AudioProcessing.decode_sp(mc, "mcep", 16000, post_filtering=True)
    assert sp1.dtype == np.float32 and sp1.shape == sp0.shape == (40, 513)
    def energy(sp):                     # mean power over the full circle from the half spectrum
        p = sp.astype(np.float64) ** 2
        return (p[:, 0] + p[:, -1] + 2 * p[:, 1:-1].sum(axis=1)) / 1024
    e0, e1 = energy(sp0), energy(sp1)
    assert np.abs(e1 / e0 - 1).max() < 1e-4                     # frame energy kept
    d0 = np.log(sp0).max(axis=1) - np.log(sp0).min(axis=1)
    d1 = np.log(sp1).max(axis=1) - np.log(sp1).min(axis=1)
    assert (d1 > d0).mean() > 0.9                               # peaks-to-valleys enhanced
