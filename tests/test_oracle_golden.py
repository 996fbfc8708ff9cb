"""Pins the CPU oracle (oracle/) against the reference's own golden vectors (CPU-only tests).

* WORLD analysis + SPTK mcep vs test/integration/fixtures/WORLD/cmp_mcep20/*.cmp (copied under
  tests/golden/): V/UV bit-exact, <= 1 float32 ulp on every other column, when run the way the
  fixtures were made (pre-emphasis 0.97, alpha 0.58, order 19 -- SURVEY.md section 4).
* MLPG vs the reference's benchmark known answer (test_AcousticModelTrainer.py:104), captured by
  tests/golden/make_golden.py with the oracle standing in for bandmat.
* the C oracle vs the numpy executable spec (oracle/world_spec.py) on a short clip.
"""
import math
import os

import numpy as np
import pytest
from scipy.io import wavfile

from idiaptts_amd.misc.utils import compute_deltas, interpolate_lin
from oracle import capi


def _read(golden_dir, name):
    fs, w = wavfile.read(os.path.join(golden_dir, name + ".wav"))
    raw = w.astype(np.float64) / 32768.0          # soundfile.read scaling (AudioProcessing.py:113)
    return np.append(raw[0], raw[1:] - 0.97 * raw[:-1]), fs   # get_raw pre-emphasis (:118)


# every utterance of the reference's fixture set (test/integration/fixtures/database/wav +
# WORLD/cmp_mcep20): byte-identical copies under tests/golden/
ALL_FIXTURES = ["LJ001-000%d" % i for i in range(1, 10)]


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_world_analysis_and_mcep_match_reference_cmp(golden_dir, name):
    x, fs = _read(golden_dir, name)
    cmp_ = np.fromfile(os.path.join(golden_dir, name + ".cmp"), dtype=np.float32).reshape(-1, 67)
    f0, sp, ap = capi.wav2world(x, fs)
    assert len(f0) == cmp_.shape[0] == capi.num_frames(len(x), fs)
    # WorldFeatLabelGen.world_extract_features (:795-805)
    lf0 = np.log(f0.clip(min=1e-10), dtype=np.float32)
    lf0[lf0 <= math.log(30)] = 0
    lf0, vuv = interpolate_lin(lf0)
    bap = np.array(capi.code_aperiodicity(ap, fs), dtype=np.float32)
    mc = capi.mcep(np.sqrt(sp), 19, 0.58).astype(np.float32)

    def within_one_ulp(a, b, ulps=1):
        # float32 units in the last place at the magnitude of these features (4.8e-7 = ulp of values
        # in [4, 8): log f0, mcep c0), or of the value itself where it is larger (bap in dB)
        a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
        tol = ulps * np.maximum(np.float32(4.8e-7), np.spacing(np.maximum(np.abs(a), np.abs(b))))
        return bool((np.abs(a - b) <= tol).all())

    assert np.array_equal(cmp_[:, 63], vuv[:, 0].astype(np.float32))          # V/UV bit-exact
    voiced = cmp_[:, 63] > 0
    assert within_one_ulp(cmp_[voiced, 60], lf0[voiced, 0])                   # lf0 of voiced frames
    # interpolated frames: float32 interpolation between end points that may each be one ulp off
    assert within_one_ulp(cmp_[~voiced, 60], lf0[~voiced, 0], ulps=2)
    assert within_one_ulp(cmp_[:, 64], bap[:, 0])                             # bap
    assert within_one_ulp(cmp_[:, :20], mc)                                   # mcep
    assert np.sqrt(np.mean((cmp_[:, :20] - mc) ** 2)) < 1e-7
    # deltas inside the cmp are np.gradient of the static columns
    d = compute_deltas(cmp_[:, :20])
    assert np.array_equal(d, cmp_[:, 20:40])
    assert np.array_equal(compute_deltas(d), cmp_[:, 40:60])


@pytest.mark.parametrize("name", ["LJ001-0002", "LJ001-0005", "LJ001-0008"])
def test_mgc2sp_is_pinned_through_the_reference_mcep(golden_dir, name):
    """pysptk.mgc2sp (AudioProcessing.py:247-256) has no golden vector, but the reference's `.cmp`
    files pin pysptk.mcep, and mcep's output is by construction the minimiser of
    E(c) = sum_w exp(R) - R - 1, R = log periodogram - log |H_c|^2, where |H_c| is exactly what
    mgc2sp(c, alpha, gamma = 0) evaluates.  So: decode the REFERENCE-HELD coefficients with the
    oracle's mgc2sp and check that they are a stationary point of E on the CheapTrick envelope the
    pinned analysis produces -- a decoder that deviated from SPTK's (wrong warping sign, missing
    c0 / 2 convention, off-by-one bin) would move the minimum away from the pinned coefficients."""
    x, fs = _read(golden_dir, name)
    cmp_ = np.fromfile(os.path.join(golden_dir, name + ".cmp"), dtype=np.float32).reshape(-1, 67)
    f0, sp, ap = capi.wav2world(x, fs)
    rows = np.arange(20, len(f0) - 20, 37)
    # coefficients at full precision: the oracle's mcep equals the .cmp to 1 float32 ulp (test above)
    mc = capi.mcep(np.sqrt(sp[rows]), 19, 0.58)
    assert np.abs(mc.astype(np.float32) - cmp_[rows, :20]).max() <= 4.8e-7
    per = sp[rows] + 1e-8                      # SPTK mcep: periodogram + eps (AudioProcessing.py:146)
    wgt = np.ones(513)
    wgt[0] = wgt[-1] = 0.5

    def criterion(c):
        r = np.log(per) - 2 * capi.mgc2sp_logamp(c, 0.58, 1024)
        return ((np.exp(r) - r - 1) * wgt).sum(axis=1)

    e0 = criterion(mc)
    # SPTK stops at a relative change of 1e-3 of E between Newton steps: the stored coefficients sit
    # within that tolerance of the exact minimiser, so a +-h step may lower E by O(h * 1e-3 * E''),
    # never by the O(h * E') a wrong decoder gives.  Measure the gradient by central differences.
    h = 1e-3
    for k in range(20):
        cp, cm = mc.copy(), mc.copy()
        cp[:, k] += h
        cm[:, k] -= h
        ep, em = criterion(cp), criterion(cm)
        grad = (ep - em) / (2 * h)
        curv = (ep - 2 * e0 + em) / (h * h)
        assert (curv > 0).all(), k                               # a minimum along every axis
        # distance to the axis minimum, in coefficient units: |grad| / curv
        assert np.abs(grad / curv).max() < 2e-3, (k, np.abs(grad / curv).max())
    # tightened: iterate the oracle's own mcep to 1e-10 and the gradient vanishes to round-off scale
    mc_t = capi.mcep(np.sqrt(sp[rows]), 19, 0.58, threshold=1e-12, maxiter=200)
    e_t = criterion(mc_t)
    for k in range(20):
        cp, cm = mc_t.copy(), mc_t.copy()
        cp[:, k] += h
        cm[:, k] -= h
        ep, em = criterion(cp), criterion(cm)
        assert np.abs((ep - em) / (2 * h) / ((ep - 2 * e_t + em) / (h * h))).max() < 2e-5, k


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_decode_aperiodicity_inverts_the_pinned_coder(golden_dir, name):
    """pyworld.decode_aperiodicity (WorldFeatLabelGen.py:933) has no golden vector; code_aperiodicity
    is pinned by the bap column of the reference's `.cmp`.  Decoding the REFERENCE-HELD bap and
    coding it again must return it: the decoder interpolates linearly (in dB) between the coarse
    band centres the coder samples, so the round trip is exact up to the clamping of unvoiced
    frames."""
    fs = 16000
    cmp_ = np.fromfile(os.path.join(golden_dir, name + ".cmp"), dtype=np.float32).reshape(-1, 67)
    bap = cmp_[:, 64:65].astype(np.float64)
    ap = capi.decode_aperiodicity(bap, fs, 1024)
    assert ap.shape == (len(bap), 513) and (ap > 0).all() and (ap <= 1.0).all()
    back = capi.code_aperiodicity(ap, fs)
    voiced = bap[:, 0] < -0.5            # WORLD's DecodeAperiodicity treats mean bap > -0.5 dB as unvoiced
    assert voiced.sum() > 50
    assert np.abs(back[voiced] - bap[voiced]).max() < 1e-9
    # unvoiced frames decode to 1 - 1e-12 and code back to (almost) 0 dB
    assert np.all(ap[~voiced] == 1.0 - 1e-12)
    assert np.abs(back[~voiced]).max() < 1e-9


def test_c_oracle_matches_numpy_spec_on_the_reference_48k_clip(golden_dir):
    """48 kHz: the reference's own wav48/p225_001.wav (test_WorldFeatLabelGen.py:611-629 runs its
    extraction on it; it stores no feature file for it).  The two independent restatements must
    agree on a clip of it: fft size 4096 for D4C / 2048 for CheapTrick, five aperiodicity bands."""
    from oracle import world_spec as ws
    fs, w = wavfile.read(os.path.join(golden_dir, "p225_001.wav"))
    assert fs == 48000
    raw = w.astype(np.float64) / 32768.0
    x = raw[19200:31200]                      # 0.25 s around the first voiced stretch (frames 80 .. 130)
    f0_c, tp = capi.dio(x, fs)
    f0_s, tp_s = ws.dio(x, fs)
    assert np.array_equal(tp, tp_s) and np.abs(f0_c - f0_s).max() < 1e-8
    f0r = capi.stonemask(x, fs, tp, f0_c)
    f0r_s = np.array([ws.stonemask_frame(x, fs, tp[i], f0_c[i]) for i in range(len(tp))])
    assert np.abs(f0r - f0r_s).max() < 1e-8
    assert (f0r > 0).sum() > 10
    fft = capi.cheaptrick_fft_size(fs)
    assert fft == 2048
    sp = capi.cheaptrick(x, fs, tp, f0r)
    floor = 3.0 * fs / (fft - 3.0)
    rng = ws.XorShift()
    for i in range(len(tp)):
        s = ws.cheaptrick_frame(x, fs, f0r[i] if f0r[i] > floor else 500.0, tp[i], fft, rng=rng)
        assert np.abs(s / sp[i] - 1).max() < 1e-9
    ap = capi.d4c(x, fs, tp, f0r)
    bap = capi.code_aperiodicity(ap, fs)
    assert bap.shape[1] == 5
    bap_s = ws.d4c_bap(x, fs, f0r, tp, fft)
    assert np.abs(bap - bap_s).max() < 1e-7


def test_normalisation_statistics_match_the_reference_held_files(golden_dir):
    """MeanCovarianceExtractor (misc/normalisation/MeanCovarianceExtractor.py:20-212) over the nine
    fixture `.cmp` files against the reference's own fixtures WORLD/cmp_mcep20/<feat>-mean-
    covariance.bin / -stats.bin (legacy layout: two int32, then float64 rows).  Those files were
    accumulated in float32 per-file sums; ours are float64, so the agreement is bounded by the
    float32 rounding of the reference (mean ~1e-6, covariance ~3e-5 absolute), and restating the
    float32 accumulation reproduces the stored mean bit for bit."""
    import struct
    from idiaptts_amd.misc.normalisation.MeanCovarianceExtractor import MeanCovarianceExtractor
    cols = {"mcep20": slice(0, 60), "lf0": slice(60, 63), "bap": slice(64, 67)}
    cmps = [np.fromfile(os.path.join(golden_dir, n + ".cmp"), dtype=np.float32).reshape(-1, 67)
            for n in ALL_FIXTURES]
    for feat, sl in cols.items():
        path = os.path.join(golden_dir, "stats", feat + "-mean-covariance.bin")
        with open(path, "rb") as f:
            n_frames, size = struct.unpack("ii", f.read(8))
            ref = np.fromfile(f, dtype=np.float64).reshape(size, -1)
        assert n_frames == sum(len(c) for c in cmps) == 11579
        ex = MeanCovarianceExtractor()
        for c in cmps:
            ex.add_sample(c[:, sl].astype(np.float64))
        mean, cov = ex.get_params()
        assert np.abs(mean - ref[0]).max() < 5e-6
        assert np.abs(cov - ref[1:]).max() < 5e-5 * max(1.0, np.abs(ref[1:]).max())
        # the loader of the legacy layout returns the stored values
        m_l, c_l, sd_l = MeanCovarianceExtractor.load(path)
        assert np.array_equal(m_l, ref[0].astype(np.float32))
        assert np.array_equal(c_l, ref[1:].astype(np.float32))
        # float32 per-file sums, as the files were made: the mean is reproduced exactly
        s32 = np.zeros(sl.stop - sl.start, np.float32)
        for c in cmps:
            s32 += c[:, sl].sum(0)
        assert np.array_equal((s32 / np.float32(n_frames)).astype(np.float64), ref[0]) or \
            np.abs(s32 / n_frames - ref[0]).max() < 1e-7


def test_mlpg_oracle_reproduces_reference_benchmark_run(golden_dir):
    kat = np.load(os.path.join(golden_dir, "mlpg_benchmark_kat.npz"))
    np.testing.assert_almost_equal((8.616, 78.4, 0.609, 37.352), kat["scores"], 3)
    from scipy.linalg import solveh_banded
    for i in range(int(kat["n_calls"])):
        feat, var, dim = kat["feat_%d" % i], kat["var_%d" % i], int(kat["dim_%d" % i])
        out = capi.mlpg(feat.astype(np.float64), var, dim)
        assert np.array_equal(out, kat["out_%d" % i])
        # independent check: LAPACK banded solve of the same normal equations, dimension 0
        T = feat.shape[0]
        tau = np.stack([np.full(T, 1.0 / var[w * dim]) for w in range(3)], 1)
        tau[0, 1:] = tau[-1, 1:] = 1e-11
        m = feat[:, [0, dim, 2 * dim]].astype(np.float64)
        bf = m * tau
        b = bf[:, 0].copy()
        b[:-1] += -0.5 * bf[1:, 1]
        b[1:] += 0.5 * bf[:-1, 1]
        b += -2.0 * bf[:, 2]
        b[:-1] += bf[1:, 2]
        b[1:] += bf[:-1, 2]
        ab = np.zeros((3, T))
        ab[2] = tau[:, 0] + 4 * tau[:, 2]
        ab[2, :-1] += 0.25 * tau[1:, 1] + tau[1:, 2]
        ab[2, 1:] += 0.25 * tau[:-1, 1] + tau[:-1, 2]
        ab[1, 1:] = -2 * (tau[:-1, 2] + tau[1:, 2])
        ab[0, 2:] = tau[1:-1, 2] - 0.25 * tau[1:-1, 1]
        ref = solveh_banded(ab, b)
        assert np.abs(ref - out[:, 0]).max() < 1e-9 * max(1.0, np.abs(ref).max())


def test_c_oracle_matches_numpy_spec_on_short_clip(golden_dir):
    """The slow numpy spec and the C oracle are independent codings of
    the same published algorithms; they must agree to fp64 round-off on 0.5 s of speech."""
    from oracle import world_spec as ws
    x, fs = _read(golden_dir, "LJ001-0008")
    x = x[4000:12000]
    f0_c, tp = capi.dio(x, fs)
    f0_s, tp_s = ws.dio(x, fs)
    assert np.array_equal(tp, tp_s) and np.abs(f0_c - f0_s).max() < 1e-8
    f0r = capi.stonemask(x, fs, tp, f0_c)
    f0r_s = np.array([ws.stonemask_frame(x, fs, tp[i], f0_c[i]) for i in range(len(tp))])
    assert np.abs(f0r - f0r_s).max() < 1e-8
    sp = capi.cheaptrick(x, fs, tp, f0r)
    fft = 1024
    floor = 3.0 * fs / (fft - 3.0)
    rng = ws.XorShift()                      # one safeguard-noise stream per CheapTrick call
    for i in range(len(tp)):
        s = ws.cheaptrick_frame(x, fs, f0r[i] if f0r[i] > floor else 500.0, tp[i], fft, rng=rng)
        assert np.abs(s / sp[i] - 1).max() < 1e-9
    ap = capi.d4c(x, fs, tp, f0r)
    bap = capi.code_aperiodicity(ap, fs)
    bap_s = ws.d4c_bap(x, fs, f0r, tp, fft)
    assert np.abs(bap - bap_s).max() < 1e-7
    mc = capi.mcep(np.sqrt(sp[::9]), 24, 0.41)
    mc_s = np.array([ws.sptk_mcep(np.sqrt(s), 24, 0.41) for s in sp[::9]])
    assert np.abs(mc - mc_s).max() < 1e-9
    la = capi.mgc2sp_logamp(mc, 0.41, fft)
    amp_s = np.array([ws.mcep_to_amp_sp(m, 0.41, fft) for m in mc])
    assert np.abs(np.exp(la) / amp_s - 1).max() < 1e-10


def test_synthesis_oracle_matches_numpy_spec_and_reference_bound(golden_dir):
    """Synthesis has no golden in the reference (parity unpinned): check the two independent
    restatements agree and that copy-synthesis passes the reference's own loose bound
    (test_WorldFeatLabelGen.py:761-763: sum (orig - resynth)^2 < 10000)."""
    from oracle import synth_spec as ss
    fs, w = wavfile.read(os.path.join(golden_dir, "LJ001-0008.wav"))
    raw = w.astype(np.float64) / 32768.0
    f0, sp, ap = capi.wav2world(raw, fs)
    y = capi.synthesize(f0, sp, ap, fs)
    assert len(y) == int(len(f0) * 5.0 * fs / 1000)
    n = min(len(y), len(raw))
    assert ((raw[:n] - y[:n]) ** 2).sum() < 10000
    # numpy spec on the first 0.6 s (it is slow); identical RNG stream => same samples
    Tn = 120
    y_c = capi.synthesize(f0[:Tn], sp[:Tn], ap[:Tn], fs)
    y_s = ss.synthesize(f0[:Tn], sp[:Tn], ap[:Tn], fs)
    assert np.abs(y_c - y_s).max() < 1e-9
    bap = capi.code_aperiodicity(ap, fs)
    ap_c = capi.decode_aperiodicity(bap, fs, 1024)
    voiced = bap.mean(1) <= -0.5
    ap_s = ss.decode_aperiodicity(bap[voiced], fs, 1024)
    assert np.abs(ap_c[voiced] - ap_s).max() < 1e-12
    assert np.all(ap_c[~voiced] == 1.0 - 1e-12)


def test_oracle_mgcep_is_the_minimiser_of_the_mgc_criterion(golden_dir):
    """pysptk.mgcep (AudioProcessing.extract_mgc, gamma = -1/3) has no golden vector in the
    reference and pysptk cannot be installed: PARITY UNPINNED.  What can be checked: (1) with
    gamma = 0 the analysis agrees with the pinned mcep restatement to the stopping tolerance of
    the two iterations; (2) for gamma < 0 its output is a stationary point of the criterion both
    methods minimise, E = sum_w exp(R) - R - 1 with R = log periodogram - log model spectrum,
    where the model spectrum is evaluated by the independent decoding path (mgc2sp: freqt,
    gnorm, gc2gc, FFT) -- no perturbation of any coefficient lowers E; (3) the reconstruction
    bound the reference's own test applies (test_WorldFeatLabelGen.py:827-836, sum of squared
    amplitude errors < 1500 per utterance) holds on the fixture's CheapTrick envelope."""
    from oracle import capi
    cmp_ = np.fromfile(os.path.join(golden_dir, "LJ001-0008.cmp"), dtype=np.float32).reshape(-1, 67)
    mc = cmp_[100:112, :20].astype(np.float64)
    amp = np.exp(capi.mgc2sp_logamp(mc, 0.58, 1024))
    x = amp ** 2 + 1e-8
    m0 = capi.mgcep(amp, 19, 0.58, 0.0)
    assert np.abs(m0 - capi.mcep(amp, 19, 0.58)).max() < 5e-3
    assert np.abs(capi.mgc2sp_gamma_logamp(mc, 0.58, 0.0, 1024)
                  - capi.mgc2sp_logamp(mc, 0.58, 1024)).max() < 1e-12

    def criterion(mgc, g):
        la = capi.mgc2sp_gamma_logamp(mgc, 0.58, g, 1024)
        r = np.log(x) - 2 * la
        w = np.ones(513)
        w[0] = w[-1] = 0.5
        return ((np.exp(r) - r - 1) * w).sum(axis=1)

    for g in (-1.0 / 3.0, -0.5):
        mgc, iters = capi.mgcep(amp, 19, 0.58, g, threshold=1e-10, maxiter=60, return_iters=True)
        assert iters.max() < 60
        e0 = criterion(mgc, g)
        for k in range(20):
            for sgn in (1.0, -1.0):
                mp = mgc.copy()
                mp[:, k] += sgn * 1e-4
                assert (criterion(mp, g) - e0).min() > -1e-9, (g, k)
        # the default stopping rule (1e-3 on log eps) lands within its tolerance of that optimum
        assert np.abs(capi.mgcep(amp, 19, 0.58, g) - mgc).max() < 2e-2
    # reference-style bound on a real envelope (CheapTrick of the fixture wav, order 59)
    from scipy.io import wavfile
    fs, w = wavfile.read(os.path.join(golden_dir, "LJ001-0008.wav"))
    raw = w[:16000].astype(np.float64) / 32768.0
    f0, sp, ap = capi.wav2world(raw, fs)
    amp_sp = np.sqrt(sp)
    alpha = 0.58
    mgc = capi.mgcep(amp_sp, 59, alpha, -1.0 / 3.0)
    rec = np.exp(capi.mgc2sp_gamma_logamp(mgc, alpha, -1.0 / 3.0, 1024).astype(np.float32))
    assert ((amp_sp - rec) ** 2).sum() < 1500 * len(amp_sp) / 600.0


def test_harvest_oracle_is_consistent_with_what_can_be_checked_here(golden_dir):
    """No reference-held vector exists for Harvest (the reference extracts with dio + stonemask and
    pyworld is not vendored): the restatement is checked against a tone of known pitch, against the
    pinned DIO + StoneMask oracle on the reference's fixture audio, and its decimation filter
    against scipy's Chebyshev design.  Parity with pyworld.harvest stays unpinned."""
    import scipy.signal
    import ctypes
    fs = 16000
    n = int(1.2 * fs)
    t = np.arange(n) / fs
    f_true = 180.0 + 40.0 * t                                  # slow glide
    phase = 2 * np.pi * np.cumsum(f_true) / fs
    x = 0.25 * sum(np.sin(k * phase) / k for k in range(1, 8))
    f0, tp = capi.harvest(x, fs)
    assert len(f0) == capi.harvest_num_frames(n, fs) == int(1000.0 * n / fs / 5.0) + 1
    mid = slice(10, len(f0) - 10)
    assert (f0[mid] > 0).all()
    truth = np.interp(tp[mid], t, f_true)
    assert np.abs(f0[mid] / truth - 1).max() < 5e-3
    # silence stays unvoiced
    assert (capi.harvest(np.zeros(4000), fs)[0] == 0).all()
    # fixture audio: where both estimators call a frame voiced they agree closely
    fs_w, w = wavfile.read(os.path.join(golden_dir, "LJ001-0008.wav"))
    xw = w.astype(np.float64) / 32768.0
    h, tph = capi.harvest(xw, fs_w)
    d, tpd = capi.dio(xw, fs_w)
    s = capi.stonemask(xw, fs_w, tpd, d)
    assert len(h) == len(s) and np.array_equal(tph, tpd)
    both = (h > 0) & (s > 0)
    assert both.sum() > 150
    assert np.median(np.abs(h[both] / s[both] - 1)) < 0.01
    # a 1 ms frame period is the estimator's own grid; 5 ms picks every fifth value of it
    h1, _ = capi.harvest(xw, fs_w, frame_period=1.0)
    assert np.array_equal(h, h1[np.minimum(len(h1) - 1, np.arange(len(h)) * 5)])
    # decimate's IIR: cheby1(3, 0.05 dB, 0.8 / r)
    fn = capi._fn("orc_decimate_coefficients", None, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p])
    for r in range(2, 13):
        a3, b2 = np.zeros(3), np.zeros(2)
        fn(r, capi._p(a3), capi._p(b2))
        b, a = scipy.signal.cheby1(3, 0.05, 0.8 / r)
        assert np.allclose(a3, -a[1:], rtol=1e-12, atol=0)
        assert np.allclose(b2, b[:2], rtol=1e-12, atol=0)
