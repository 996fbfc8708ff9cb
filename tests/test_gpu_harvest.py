"""GPU parity of the Harvest F0 kernels (through the C ABI) against the C restatement of WORLD's
harvest.cpp in oracle/c/harvest.c, stage by stage and end to end.

The oracle convolves the 152 band-pass filters through 2^16-point FFTs and refines every candidate
with two FFTs; the kernels evaluate the same sums directly (matrix-core FIR, six spectral bins per
candidate), so values agree to rounding (~1e-10 relative) rather than bit for bit; the voicing
decisions built on them must agree exactly.  The spectrum bins WORLD overwrites while it multiplies
(GetFilteredSignal's mirrored write, ~1e-5 of the band level in the short filters) are part of the
oracle and of the kernels."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

pytestmark = pytest.mark.gpu


def _read(golden_dir, name):
    fs, w = wavfile.read(os.path.join(golden_dir, name + ".wav"))
    return w.astype(np.float64) / 32768.0, fs


def _synthetic(fs, seconds, seed):
    rng = np.random.default_rng(4321 + seed)
    n = int(fs * seconds)
    f0 = np.clip(140 + 30 * seed + np.cumsum(rng.normal(0, 0.02, n)) * 20, 80, 400)
    voiced = (np.sin(2 * np.pi * np.arange(n) / fs * 1.7 + seed) > -0.2).astype(float)
    phase = 2 * np.pi * np.cumsum(f0) / fs
    src = sum(np.sin(k * phase) / k for k in range(1, 10)) * voiced
    return 0.3 * src / np.abs(src).max() + 10 ** (-45 / 20) * rng.normal(size=n)


def _run(gpu, xs, fs, frame_period=5.0, **kw):
    from idiaptts_amd import ops
    x_off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).tolist()
    T = [ops.harvest_num_frames(len(x), fs, frame_period) for x in xs]
    f_off = np.concatenate([[0], np.cumsum(T)]).tolist()
    x = torch.from_numpy(np.concatenate(xs)).to(gpu)
    out = ops.harvest(x, x_off, f_off, fs, frame_period, **kw)
    return out, f_off


def _close(a, b, rtol):
    """same voicing pattern, voiced values within rtol"""
    assert a.shape == b.shape
    assert np.array_equal(a > 0, b > 0), "voicing differs at {}".format(np.nonzero((a > 0) != (b > 0))[0][:10])
    m = b > 0
    if m.any():
        assert np.abs(a[m] / b[m] - 1).max() < rtol, np.abs(a[m] / b[m] - 1).max()


@pytest.mark.parametrize("name", ["LJ001-0008", "LJ001-0002"])
def test_every_stage_matches_the_oracle_on_the_fixture_audio(gpu, golden_dir, name):
    from oracle import capi
    x, fs = _read(golden_dir, name)
    f0_ref, tp, ref = capi.harvest(x, fs, debug=True)
    (f0, dbg), _ = _run(gpu, [x], fs, stages=True)
    nc = ref["n_cand"]
    raw = dbg["raw"].cpu().numpy()
    _close(raw, ref["raw"], 1e-9)
    cand = dbg["cand"].cpu().numpy()[:, :nc]
    score = dbg["score"].cpu().numpy()[:, :nc]
    _close(cand, ref["cand"][:, :nc], 1e-8)
    m = ref["score"][:, :nc] > 0
    assert np.abs(score[m] / ref["score"][:, :nc][m] - 1).max() < 1e-6
    _close(dbg["best"].cpu().numpy(), ref["best"], 1e-8)
    f0 = f0.cpu().numpy()
    assert len(f0) == len(f0_ref) == capi.harvest_num_frames(len(x), fs)
    _close(f0, f0_ref, 1e-8)
    assert (f0 > 0).sum() > 50


def test_ragged_batch_other_rates_and_frame_periods(gpu, golden_dir):
    from oracle import capi
    for fs, fp, kw in [(16000, 5.0, {}), (22050, 5.0, {}), (48000, 10.0, {}), (8000, 1.0, {}),
                       (16000, 5.0, dict(f0_floor=60.0, f0_ceil=500.0))]:
        xs = [_synthetic(fs, 0.35 + 0.4 * k, k) for k in range(3)]
        f0, f_off = _run(gpu, xs, fs, fp, **kw)
        f0 = f0.cpu().numpy()
        for u, x in enumerate(xs):
            ref, _ = capi.harvest(x, fs, fp, **kw)
            _close(f0[f_off[u]:f_off[u + 1]], ref, 1e-7)


def test_silence_and_noise_are_unvoiced_like_the_oracle(gpu):
    from oracle import capi
    fs = 16000
    rng = np.random.default_rng(5)
    xs = [np.zeros(4000), 1e-3 * rng.normal(size=6000), _synthetic(fs, 0.5, 1)]
    f0, f_off = _run(gpu, xs, fs)
    f0 = f0.cpu().numpy()
    for u, x in enumerate(xs):
        ref, _ = capi.harvest(x, fs)
        _close(f0[f_off[u]:f_off[u + 1]], ref, 1e-7)
    assert (f0[:f_off[1]] == 0).all()


def test_batched_call_equals_one_call_per_utterance(gpu, golden_dir):
    xs = [_read(golden_dir, n)[0] for n in ("LJ001-0008", "LJ001-0002")] + [_synthetic(16000, 1.0, 2)]
    f0, f_off = _run(gpu, xs, 16000)
    for u, x in enumerate(xs):
        single, _ = _run(gpu, [x], 16000)
        assert torch.equal(single, f0[f_off[u]:f_off[u + 1]])


def test_the_extractor_switch_routes_f0_through_harvest(gpu, golden_dir):
    """WorldFeatLabelGen.f0_estimator = "harvest": lf0 / vuv come from Harvest, and the envelope
    and aperiodicity are analysed at Harvest's F0 (pyworld.cheaptrick / d4c after pyworld.harvest)."""
    from idiaptts_amd import world
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    from oracle import capi
    x, fs = _read(golden_dir, "LJ001-0008")
    before = WorldFeatLabelGen.f0_estimator
    try:
        WorldFeatLabelGen.f0_estimator = "harvest"
        amp_sp, lf0, vuv, bap = WorldFeatLabelGen.world_extract_features(x, fs, 5)
    finally:
        WorldFeatLabelGen.f0_estimator = before
    ref, tp = capi.harvest(x, fs)
    lf0_ref, vuv_ref = world.lf0_vuv_from_f0(ref)
    assert np.array_equal(vuv, vuv_ref)
    assert np.abs(lf0 - lf0_ref).max() < 1e-5
    sp_ref = capi.cheaptrick(x, fs, tp, ref)
    assert np.abs(amp_sp ** 2 / sp_ref - 1).max() < 1e-6
    with pytest.raises(NotImplementedError):
        world.analyse_batch([x], fs, f0_method="swipe")


def test_tiny_and_long_utterances(gpu):
    """utterances of a few frames, and one long enough that its event lists and section store are
    sized by the general formulas rather than by the fixtures"""
    from oracle import capi
    fs = 16000
    rng = np.random.default_rng(11)
    xs = [1e-2 * rng.normal(size=n) for n in (64, 161, 800)] + [_synthetic(fs, 21.0, 3)]
    f0, f_off = _run(gpu, xs, fs)
    f0 = f0.cpu().numpy()
    for u, x in enumerate(xs):
        ref, _ = capi.harvest(x, fs)
        _close(f0[f_off[u]:f_off[u + 1]], ref, 1e-7)


def test_a_batch_larger_than_the_scratch_budget_is_processed_in_chunks(gpu):
    """~130 MB of scratch per 6 s utterance against a 12 GB budget: 112 utterances need two
    sub-batches; every utterance must come out as if it had been analysed alone"""
    from oracle import capi
    fs = 16000
    xs = [_synthetic(fs, 5.5 + 0.01 * k, k % 7) for k in range(112)]
    f0, f_off = _run(gpu, xs, fs)
    for u in (0, 57, 111):
        single, _ = _run(gpu, [xs[u]], fs)
        assert torch.equal(single, f0[f_off[u]:f_off[u + 1]])
    ref, _ = capi.harvest(xs[111], fs)
    _close(f0[f_off[111]:f_off[112]].cpu().numpy(), ref, 1e-7)


def test_failed_calls_hand_their_scratch_back(gpu):
    """An entry point that returns early (bad frame offsets are detected after the first scratch
    blocks have been taken) must not leave those blocks marked busy: a data-preparation loop that
    catches the error and goes on would otherwise leak once per bad file.  The blocks are released
    by the entry point's ScratchScope."""
    import ctypes
    from idiaptts_amd import lib as _lib, ops
    L = _lib.load()

    def used():
        r, u, k = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(L.itts_scratch_pool_stats(ctypes.byref(r), ctypes.byref(u), ctypes.byref(k)), "stats")
        return u.value

    fs = 16000
    x = torch.from_numpy(np.random.default_rng(0).normal(size=fs) * 0.1).to(gpu)
    T = ops.harvest_num_frames(fs, fs)
    ops.harvest(x, [0, fs], [0, T], fs)                  # a good call: everything handed back
    torch.cuda.synchronize()
    base = used()
    for _ in range(5):
        with pytest.raises(_lib.IttsError):
            ops.harvest(x, [0, fs], [0, T + 3], fs)      # frame offsets do not match the frame count
        with pytest.raises(_lib.IttsError):
            ops.dio(x, [0, fs], [0, T + 3], fs)
    torch.cuda.synchronize()
    assert used() == base
    f0 = ops.harvest(x, [0, fs], [0, T], fs)             # and the library still works
    assert f0.shape[0] == T
