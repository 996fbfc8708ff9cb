"""Size-independent properties at the sizes of the bench / BASELINE configs (where the CPU oracle
would take minutes): batch independence, determinism, linearity, round trips."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_world_batch_of_48_utterances_equals_utterance_by_utterance(gpu):
    """Config 5 batch (48 utterances, ~290 s, 58 k frames): every kernel of the analysis /
    synthesis chain treats utterances independently, so the rows of an utterance inside the batch
    are bit-identical to the result of analysing it alone."""
    from idiaptts_amd import ops, world
    from idiaptts_amd.bench_support import make_audio_batch
    fs = 16000
    raws = make_audio_batch(48, fs, seed=0)
    x_off = world.offsets([len(r) for r in raws])
    f_off = world.offsets([world.num_frames(len(r), fs, 5.0) for r in raws])
    x = torch.from_numpy(np.concatenate(raws)).to(gpu)

    def chain(x, x_off, f_off):
        f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs), f_off, fs)
        sp, mc, it = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, order=59, alpha=0.41,
                                         mc_dtype=torch.float64, want_iters=True)
        ap, bap = ops.d4c(x, x_off, f0, f_off, fs, want_bap=torch.float64)
        y, y_off = ops.world_synthesize(f0, sp, ap, f_off, fs, dtype=torch.float64)
        return f0, sp, mc, it, ap, bap, y, y_off

    f0, sp, mc, it, ap, bap, y, y_off = chain(x, x_off, f_off)
    assert f_off[-1] > 55000 and torch.isfinite(mc).all() and torch.isfinite(y).all()
    for u in (0, 17, 47):
        xu = torch.from_numpy(raws[u]).to(gpu)
        T = f_off[u + 1] - f_off[u]
        f0u, spu, mcu, itu, apu, bapu, yu, _ = chain(xu, [0, len(raws[u])], [0, T])
        a, b = f_off[u], f_off[u + 1]
        for whole, alone in ((f0, f0u), (sp, spu), (mc, mcu), (it, itu), (ap, apu), (bap, bapu)):
            assert torch.equal(whole[a:b], alone)
        # (overlap-add uses f64 atomics: the order of the ~10 pulses meeting in a sample is not
        # fixed by the programming model, so the waveform is compared to round-off, not to the bit)
        assert (y[y_off[u]:y_off[u + 1]] - yu).abs().max() < 1e-13
    # re-analysis of the synthesised batch finds (nearly) the same voicing and pitch
    f_off2 = world.offsets([world.num_frames(y_off[u + 1] - y_off[u], fs, 5.0) for u in range(48)])
    f0r_all = ops.stonemask(y, y_off, ops.dio(y, y_off, f_off2, fs), f_off2, fs)
    # the synthesised signal is T * 80 samples long: it has the same number of frames or one more
    f0r = torch.cat([f0r_all[f_off2[u]:f_off2[u] + (f_off[u + 1] - f_off[u])] for u in range(48)])
    v0, v1 = f0 > 0, f0r > 0
    assert float((v0 == v1).double().mean()) > 0.9          # boundary frames flip, nothing else
    both = v0 & v1
    cents = 1200 * torch.log2(f0r[both] / f0[both])
    assert float(cents.abs().median()) < 5.0


def test_mlpg_is_linear_in_the_means_at_config_4_size(gpu):
    """256 utterances, 310 k frames x 62 dims: P x = b with b linear in the means, so
    mlpg(a f + c g) = a mlpg(f) + c mlpg(g) for fixed variances (to fp64 round-off), and a
    constant static trajectory with zero deltas is reproduced."""
    from idiaptts_amd import ops, world
    from idiaptts_amd.bench_support import utterance_lengths
    off = world.offsets(utterance_lengths(256, seed=5).tolist())
    n = off[-1]
    g = torch.Generator(device="cpu").manual_seed(0)
    f = torch.randn(n, 186, dtype=torch.float64, generator=g).to(gpu)
    h = torch.randn(n, 186, dtype=torch.float64, generator=g).to(gpu)
    var = (torch.rand(186, dtype=torch.float64, generator=g) * 0.99 + 0.01).to(gpu)
    mf, mh = ops.mlpg_generation(f, var, 62, off), ops.mlpg_generation(h, var, 62, off)
    mix = ops.mlpg_generation(0.3 * f - 1.7 * h, var, 62, off)
    assert (mix - (0.3 * mf - 1.7 * mh)).abs().max() < 1e-10 * max(1.0, float(mf.abs().max()))
    const = torch.zeros(n, 186, dtype=torch.float64, device=gpu)
    const[:, :62] = torch.linspace(-2, 2, 62, dtype=torch.float64, device=gpu)
    out = ops.mlpg_generation(const, var, 62, off)
    assert (out - const[:, :62]).abs().max() < 1e-9


def test_ff_train_step_is_deterministic_at_bench_size(gpu):
    """Config 2 step (32 utterances, 39 k frames): split-K slabs are reduced in a fixed order, so
    two runs from the same state give the same bits in the loss and in every parameter."""
    from idiaptts_amd.bench_support import make_ff_batch
    from idiaptts_amd.native_ff import FlatFFModel
    x, y, lengths = make_ff_batch(32, seed=0, device=gpu)
    valid = torch.ones(x.shape[0], dtype=torch.uint8, device=gpu)
    runs = []
    for _ in range(2):
        model = FlatFFModel((425, 512, 512, 187), ("tanh", "tanh", None), device=gpu, seed=0)
        losses = [float(model.train_step(x, y, valid, float(lengths.sum()))) for _ in range(5)]
        runs.append((losses, model.params.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0][-1] < runs[0][0][0]


def test_bilstm_output_does_not_depend_on_the_order_of_the_batch(gpu):
    """Config 3 shape (64 utterances, 3 x 512 BiLSTM): rows are sorted by length internally;
    permuting the utterances permutes the outputs and nothing else (bit for bit), padding stays
    zero."""
    from idiaptts_amd.nn import LSTM
    torch.manual_seed(0)
    B, H = 64, 512
    lengths = torch.randint(40, 200, (B,))
    T = int(lengths.max())
    x = torch.randn(T, B, 425, device=gpu)
    for b in range(B):
        x[lengths[b]:, b] = 0
    net = LSTM(425, H, 3, bidirectional=True).to(gpu)
    with torch.no_grad():
        out, _ = net(x, None, lengths)
        perm = torch.randperm(B)
        out_p, _ = net(x[:, perm], None, lengths[perm])
    assert torch.equal(out[:, perm], out_p)
    for b in (0, 31, 63):
        assert float(out[lengths[b]:, b].abs().max()) == 0.0 if lengths[b] < T else True


ALL_FIXTURES = ["LJ001-000%d" % i for i in range(1, 10)]


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_closed_loop_through_the_reference_held_features(gpu, golden_dir, name):
    """Synthesis-side parity cannot be pinned to a reference waveform (the reference holds none),
    but the loop can be closed through the PINNED half: the reference's golden `.cmp` features
    (mcep20 / lf0 / V-UV / bap) -> decode_sp -> world_features_to_raw (HIP synthesis with
    de-pre-emphasis) -> get_raw-style pre-emphasis -> HIP analysis as the fixtures were made
    (alpha 0.58, order 19) -> compared with the `.cmp` again, next to the same loop through the C
    oracle.  Measured for the oracle loop on these fixtures: V/UV agreement 0.70 - 0.74 (one-sided: the
    fixture contours hold runs of 390-465 Hz octave-jump frames labelled voiced, which a
    re-analysis of the clean resynthesis calls unvoiced; 1 - 18 frames per utterance go the other way), lf0
    RMSE on commonly voiced frames 24-35 cents (without the 0 - 5.3 % octave-type errors), MCD 3.1-3.3 dB (order-19 envelope re-estimated from
    its own resynthesis).  The HIP loop must reproduce the oracle loop, and both must stay inside
    those figures."""
    import scipy.signal
    from idiaptts_amd.src.data_preparation.audio.AudioProcessing import AudioProcessing
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    from oracle import capi
    fs, alpha, pre = 16000, 0.58, 0.97
    cmp_ = np.fromfile(os.path.join(golden_dir, name + ".cmp"), dtype=np.float32).reshape(-1, 67)
    T = len(cmp_)
    mc, lf0, vuv, bap = cmp_[:, :20], cmp_[:, 60], cmp_[:, 63], cmp_[:, 64:65]

    def metrics(mc_b, f0_b):
        vb = (f0_b > 0).astype(np.float32)
        both = (vb == 1) & (vuv == 1)
        cents = (np.log(f0_b[both]) - lf0[both].astype(np.float64)) * 1200 / np.log(2)
        gross = np.abs(cents) > 300          # octave-type errors: counted, not averaged
        lf0_rmse = np.sqrt(np.mean(cents[~gross] ** 2))
        mcd = (10 / np.log(10)) * np.sqrt(2 * ((mc_b[:, 1:] - mc[:, 1:].astype(np.float64)) ** 2)
                                          .sum(1)).mean()
        return vb, float((vb == vuv).mean()), int(((vuv == 0) & (vb == 1)).sum()), \
            lf0_rmse, mcd, float(gross.mean())

    # HIP loop through the drop-in API
    amp_sp = AudioProcessing.decode_sp(mc.astype(np.float64), "mcep", fs, alpha)
    wav = WorldFeatLabelGen.world_features_to_raw(amp_sp, lf0.copy(), vuv.copy(), bap.copy(), fs,
                                                  n_fft=1024, preemphasis=pre)
    assert len(wav) == T * 80
    raw = np.append(wav[0], wav[1:] - pre * wav[:-1])
    res = WorldFeatLabelGen.extract_features_batch([raw], fs, num_coded_sps=20, mgc_alpha=alpha)[0]
    mc_h, lf0_h, vuv_h = res[0][:T].astype(np.float64), res[1][:T, 0], res[2][:T, 0]
    f0_h = np.exp(lf0_h.astype(np.float64)) * vuv_h
    vb_h, agree_h, extra_h, cents_h, mcd_h, gross_h = metrics(mc_h, f0_h)
    # the same loop through the C oracle
    pw = np.exp(capi.mgc2sp_logamp(mc.astype(np.float64), alpha, 1024).astype(np.float32)) \
        .astype(np.float64) ** 2
    f0 = np.exp(lf0.astype(np.float64))
    v = vuv.copy()
    v[f0 < 30] = 0
    f0[v == 0] = 0
    y = capi.synthesize(f0, pw, capi.decode_aperiodicity(bap.astype(np.float64), fs, 1024), fs)
    y = scipy.signal.lfilter([1], [1, -pre], y.astype(np.float32))
    raw_o = np.append(y[0], y[1:] - pre * y[:-1])
    f0_o, sp_o, _ = capi.wav2world(raw_o, fs)
    mc_o = capi.mcep(np.sqrt(sp_o), 19, alpha)[:T]
    vb_o, agree_o, extra_o, cents_o, mcd_o, gross_o = metrics(mc_o, f0_o[:T])
    # HIP loop == oracle loop
    assert np.sqrt(np.mean((wav - y) ** 2)) < 1e-6
    assert (vb_h == vb_o).mean() >= 0.995
    assert abs(cents_h - cents_o) < 0.5 and abs(mcd_h - mcd_o) < 0.02 and abs(gross_h - gross_o) < 0.004
    # both inside what the fixtures allow
    for agree, extra, cents, mcd, gross in ((agree_h, extra_h, cents_h, mcd_h, gross_h),
                                            (agree_o, extra_o, cents_o, mcd_o, gross_o)):
        assert agree > 0.65 and extra <= max(3, 0.02 * T)   # (0.6 - 1.8 % of the frames on the nine fixtures)
        # lf0: 24 - 35 cents RMSE over the commonly voiced frames, leaving out the 0 - 5.3 % of them
        # that are more than 300 cents off (octave-type errors of the re-analysis, LJ001-0005: 29 of 546)
        assert cents < 40.0 and gross <= 0.06 and mcd < 4.0


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_copy_synthesis_bound_of_the_reference_on_every_fixture_wav(gpu, golden_dir, name):
    """test_WorldFeatLabelGen.py:761-763: sum (original - WORLD resynthesis)^2 < 10000, through
    the HIP analysis + synthesis, for every committed fixture wav."""
    from idiaptts_amd.src.data_preparation.audio.AudioProcessing import AudioProcessing
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    raw, fs = AudioProcessing.get_raw(os.path.join(golden_dir, name + ".wav"), 0.0)
    amp_sp, lf0, vuv, bap = WorldFeatLabelGen.world_extract_features(raw, fs, 5)
    wav = WorldFeatLabelGen.world_features_to_raw(amp_sp, lf0, vuv, bap, fs=fs, n_fft=1024)
    n = min(len(wav), len(raw))
    assert ((raw[:n] - wav[:n]) ** 2).sum() < 10000


def test_harvest_is_invariant_to_the_signal_level_at_bench_size(gpu):
    """Every stage of Harvest is homogeneous in the signal (decimation, DC removal, band-pass
    filters, zero-crossing positions, instantaneous frequencies are ratios): scaling the audio by a
    power of two scales every intermediate exactly, so the contour may only move through the two
    1e-12 safeguards of the refinement.  64 utterances of 2-10 s (bench.py's Harvest workload)."""
    from idiaptts_amd import ops
    from idiaptts_amd.bench_support import make_audio_batch
    fs = 16000
    raws = make_audio_batch(64, fs, seed=2)
    x_off = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).tolist()
    f_off = np.concatenate([[0], np.cumsum([ops.harvest_num_frames(len(r), fs, 5.0) for r in raws])]).tolist()
    x = torch.from_numpy(np.concatenate(raws)).to(gpu)
    a = ops.harvest(x, x_off, f_off, fs)
    b = ops.harvest(x * 0.25, x_off, f_off, fs)
    va, vb = a > 0, b > 0
    assert torch.equal(va, vb)
    assert float(((a[va] - b[va]).abs() / a[va]).max()) < 1e-8
    # in range, or unvoiced
    assert bool(((a[va] >= 71.0) & (a[va] <= 800.0)).all())
    assert 0.2 < float(va.double().mean()) < 0.8
