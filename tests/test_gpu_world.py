"""GPU parity of the WORLD / SPTK kernels (through the C ABI) against the C oracle, on the
reference's fixture audio (tests/golden) and on synthetic audio. North-star bars: bit-exact
V/UV and frame counts, <= 1e-4 RMSE on MGC / BAP / LF0 (we assert far tighter bounds)."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

pytestmark = pytest.mark.gpu


def _read(golden_dir, name, pre=0.97):
    fs, w = wavfile.read(os.path.join(golden_dir, name + ".wav"))
    raw = w.astype(np.float64) / 32768.0
    return np.append(raw[0], raw[1:] - pre * raw[:-1]), fs


def _synthetic(fs, seconds, seed):
    """SURVEY.md section 8d style signal: harmonic source with F0 random walk, AR-shaped, + noise."""
    rng = np.random.default_rng(1234 + seed)
    n = int(fs * seconds)
    f0 = np.clip(150 + np.cumsum(rng.normal(0, 0.02, n)) * 20, 90, 300)
    voiced = (np.sin(2 * np.pi * np.arange(n) / fs * 1.3 + seed) > -0.3).astype(float)
    phase = 2 * np.pi * np.cumsum(f0) / fs
    src = sum(np.sin(k * phase) / k for k in range(1, 12)) * voiced
    x = 0.3 * src / np.abs(src).max() + 10 ** (-40 / 20) * rng.normal(size=n)
    return x


@pytest.fixture(scope="module")
def utts(golden_dir):
    from oracle import capi
    out = []
    for name in ["LJ001-0008", "LJ001-0002"]:
        x, fs = _read(golden_dir, name)
        f0, tp = capi.dio(x, fs)
        f0 = capi.stonemask(x, fs, tp, f0)
        out.append((x, fs, f0, tp))
    return out


def _batch(utts, gpu):
    xs = np.concatenate([u[0] for u in utts])
    f0 = np.concatenate([u[2] for u in utts])
    x_off = np.concatenate([[0], np.cumsum([len(u[0]) for u in utts])]).tolist()
    f_off = np.concatenate([[0], np.cumsum([len(u[2]) for u in utts])]).tolist()
    return torch.from_numpy(xs).to(gpu), torch.from_numpy(f0).to(gpu), x_off, f_off


def test_cheaptrick_and_fused_mcep_match_oracle(gpu, utts, golden_dir):
    from idiaptts_amd import ops
    from oracle import capi
    x, f0, x_off, f_off = _batch(utts, gpu)
    fs = utts[0][1]
    sp, mc, iters = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, order=19, alpha=0.58,
                                        mc_dtype=torch.float64, want_iters=True)
    sp = sp.cpu().numpy()
    mc = mc.cpu().numpy()
    iters = iters.cpu().numpy()
    for u, (xu, _, f0u, tpu) in enumerate(utts):
        a, b = f_off[u], f_off[u + 1]
        sp_ref = capi.cheaptrick(xu, fs, tpu, f0u)
        # The smoothing differences running sums over the whole spectrum: a bin 1e-7 below its frame's
        # peak carries the rounding of the sums, eps * 1e7 in relative terms, in the oracle (sequential
        # sum) as in the kernels (blocked sums).  scripts/ct_error_probe.py: the worst bin of this
        # fixture is 1.2e-8 off for the wave kernel AND for the workgroup kernel (at fft 2048), at
        # sp = 3e-10 against a frame peak of 6e-3; median 1e-14, 99 % below 2e-10, 99.99 % below 7e-9.
        # So: the relative bound with room for that worst bin, a tight one for the bulk, and every bin
        # within 1e-11 of its frame's peak.
        err = np.abs(sp[a:b] - sp_ref)
        peak = sp_ref.max(axis=1, keepdims=True)
        assert (err / sp_ref).max() < 3e-8
        assert np.quantile(err / sp_ref, 0.999) < 2e-9
        assert (err / peak).max() < 1e-11
        mc_ref, it_ref = capi.mcep(np.sqrt(sp_ref), 19, 0.58, return_iters=True)
        assert np.array_equal(iters[a:b], it_ref)            # same Newton trip counts
        assert np.abs(mc[a:b] - mc_ref).max() < 1e-8
    # and against the reference's golden cmp (float32 mcep columns): <= 1 ulp
    cmp_ = np.fromfile(os.path.join(golden_dir, "LJ001-0008.cmp"), dtype=np.float32).reshape(-1, 67)
    n0 = f_off[1]
    assert np.abs(mc[:n0].astype(np.float32) - cmp_[:, :20]).max() <= 4.8e-7


def test_randn_stream_through_the_lds_tile_is_the_plain_kernels_stream(gpu, utts):
    """csrc/context.hip, randn_u32_kernel: the lanes' chunks of normals leave through a transposing LDS tile
    (128-byte runs) instead of 64 stores 256 bytes apart.  CheapTrick adds the stream * 1e-12 to every windowed
    segment, five orders of magnitude above an ulp of the samples: one wrong, missing or shifted normal changes the
    bits of the spectrum.  Ragged lengths put the end of the stream inside a chunk, a half chunk and a wave."""
    from idiaptts_amd import ops
    fs = utts[0][1]
    def cut(u, n):
        return (u[0][:n], fs, u[2][:int(1000.0 * n / fs / 5.0) + 1], None)
    cases = [utts, [cut(u, len(u[0]) // 3) for u in utts[:2]], [cut(utts[0], 4000), cut(utts[1], 900)]]
    old = os.environ.get("ITTS_RANDN_DIRECT")
    try:
        for case in cases:
            x, f0, x_off, f_off = _batch(case, gpu)
            res = {}
            for mode in ("1", "0"):
                os.environ["ITTS_RANDN_DIRECT"] = mode
                sp, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, order=19, alpha=0.58, mc_dtype=torch.float64)[:2]
                res[mode] = sp.cpu().numpy()
            assert np.array_equal(res["0"], res["1"])
    finally:
        if old is None:
            os.environ.pop("ITTS_RANDN_DIRECT", None)
        else:
            os.environ["ITTS_RANDN_DIRECT"] = old


@pytest.mark.parametrize("order", [19, 24, 59])
def test_fused_newton_products_equal_the_two_launches_bit_for_bit(gpu, utts, order):
    """csrc/mcep_lockstep.hip, mcls_fused3_kernel (round 5): a Newton round's two
    products in one kernel -- the [frames x 513] ratio never leaves the CU -- against the two launches it replaces
    (ITTS_MCEP_FUSED=0): same K permutation, same order of accumulation, same epilogue expression, so the
    mel-cepstra are the same bits and the trip counts the same numbers (orders 19 / 24: cr is up to 64 wide, 59: up
    to 128; 79 has no fused form)."""
    from idiaptts_amd import ops
    from oracle import capi
    amp = torch.from_numpy(np.concatenate([np.sqrt(capi.cheaptrick(xu, fs, tpu, f0u)) for xu, fs, f0u, tpu in utts[:2]])).to(gpu)
    res = {}
    old = os.environ.get("ITTS_MCEP_FUSED")
    try:
        for mode in ("0", "1"):
            os.environ["ITTS_MCEP_FUSED"] = mode
            mc, iters = ops.mcep(amp, order, 0.41, dtype=torch.float64, want_iters=True)
            res[mode] = (mc.cpu().numpy(), iters.cpu().numpy())
    finally:
        if old is None:
            os.environ.pop("ITTS_MCEP_FUSED", None)
        else:
            os.environ["ITTS_MCEP_FUSED"] = old
    assert np.array_equal(res["0"][1], res["1"][1])
    assert np.array_equal(res["0"][0], res["1"][0])
    assert res["1"][1].max() > 2            # the loop did iterate


@pytest.mark.parametrize("n_frames", [1, 2, 15, 16, 17, 127, 128, 129, 300])
def test_fused_newton_products_on_ragged_frame_counts(gpu, utts, n_frames):
    """The fused kernels own 16 frames per wave and 128 per workgroup: frame counts around those sizes (a lone frame,
    one short of / one past a wave and a workgroup) against the two-launch form, bit for bit -- rows past the end
    repeat the last frame inside the kernels and must not be stored."""
    from idiaptts_amd import ops
    from oracle import capi
    xu, fs, f0u, tpu = utts[0]
    amp_all = np.sqrt(capi.cheaptrick(xu, fs, tpu, f0u))
    amp = torch.from_numpy(np.ascontiguousarray(amp_all[40:40 + n_frames])).to(gpu)
    guard = torch.full((n_frames + 8, 60), 7.25, dtype=torch.float64, device=gpu)      # nothing may be written past the rows
    res = {}
    old = os.environ.get("ITTS_MCEP_FUSED")
    try:
        for mode in ("0", "1"):
            os.environ["ITTS_MCEP_FUSED"] = mode
            mc, iters = ops.mcep(amp, 59, 0.41, dtype=torch.float64, want_iters=True)
            res[mode] = (mc.cpu().numpy(), iters.cpu().numpy())
    finally:
        if old is None:
            os.environ.pop("ITTS_MCEP_FUSED", None)
        else:
            os.environ["ITTS_MCEP_FUSED"] = old
    assert res["1"][0].shape == (n_frames, 60) and np.isfinite(res["1"][0]).all()
    assert np.array_equal(res["0"][1], res["1"][1])
    assert np.array_equal(res["0"][0], res["1"][0])
    mc_ref, it_ref = capi.mcep(amp_all[40:40 + n_frames], 59, 0.41, return_iters=True)
    assert np.array_equal(res["1"][1], it_ref)
    assert np.abs(res["1"][0] - mc_ref).max() < 1e-8
    assert float(guard.min()) == 7.25


@pytest.mark.parametrize("order,alpha", [(59, 0.41), (24, 0.41), (79, 0.58)])
def test_mcep_from_amp_and_mgc2sp_roundtrip(gpu, utts, order, alpha):
    from idiaptts_amd import ops
    from oracle import capi
    xu, fs, f0u, tpu = utts[0]
    sp_ref = capi.cheaptrick(xu, fs, tpu, f0u)
    amp = np.sqrt(sp_ref)
    mc, iters = ops.mcep(torch.from_numpy(amp).to(gpu), order, alpha, dtype=torch.float64,
                         want_iters=True)
    mc_ref, it_ref = capi.mcep(amp, order, alpha, return_iters=True)
    assert np.array_equal(iters.cpu().numpy(), it_ref)
    assert np.sqrt(np.mean((mc.cpu().numpy() - mc_ref) ** 2)) < 1e-8
    mc32 = ops.mcep(torch.from_numpy(amp).to(gpu), order, alpha)
    assert mc32.dtype == torch.float32
    assert np.abs(mc32.cpu().numpy() - mc_ref.astype(np.float32)).max() <= 1e-6
    # mgc2sp: log amplitude vs oracle and the reference's float32 exp form
    la = ops.mgc2sp(mc, alpha, 1024, want_logamp=True).cpu().numpy()
    la_ref = capi.mgc2sp_logamp(mc.cpu().numpy(), alpha, 1024)
    assert np.abs(la - la_ref).max() < 1e-10
    amp32 = ops.mgc2sp(mc, alpha, 1024).cpu().numpy()
    ref32 = np.exp(la_ref.astype(np.float32))
    assert np.abs(amp32 / ref32 - 1).max() < 1e-6
    if order == 79:
        # reference bound: sum (amp - reconstruction)^2 < 100 (test_WorldFeatLabelGen.py:816-824)
        assert ((amp - amp32) ** 2).sum() < 100


def test_code_decode_aperiodicity(gpu, utts):
    from idiaptts_amd import ops
    from oracle import capi
    xu, fs, f0u, tpu = utts[0]
    ap = capi.d4c(xu, fs, tpu, f0u)
    bap_ref = capi.code_aperiodicity(ap, fs)
    bap = ops.code_aperiodicity(torch.from_numpy(ap).to(gpu), fs).cpu().numpy()
    assert np.abs(bap - bap_ref).max() < 1e-10
    bap32 = ops.code_aperiodicity(torch.from_numpy(ap).to(gpu), fs, dtype=torch.float32)
    assert np.array_equal(bap32.cpu().numpy(), bap_ref.astype(np.float32))
    dec_ref = capi.decode_aperiodicity(bap_ref, fs, 1024)
    dec = ops.decode_aperiodicity(torch.from_numpy(bap_ref).to(gpu), fs, 1024).cpu().numpy()
    assert np.abs(dec - dec_ref).max() < 1e-12
    # 48 kHz: 5 bands, fft 2048
    rng = np.random.default_rng(0)
    bap5 = -rng.uniform(0.0, 30.0, size=(50, 5))
    bap5[::7] = -1e-12
    d5 = ops.decode_aperiodicity(torch.from_numpy(bap5).to(gpu), 48000, 2048).cpu().numpy()
    assert np.abs(d5 - capi.decode_aperiodicity(bap5, 48000, 2048)).max() < 1e-12
    c5 = ops.code_aperiodicity(torch.from_numpy(d5).to(gpu), 48000).cpu().numpy()
    assert np.abs(c5 - capi.code_aperiodicity(d5, 48000)).max() < 1e-9


def test_cheaptrick_48k_synthetic(gpu):
    from idiaptts_amd import ops
    from oracle import capi
    fs = 48000
    x = _synthetic(fs, 0.8, 1)
    f0, tp = capi.dio(x, fs)
    f0 = capi.stonemask(x, fs, tp, f0)
    sp, mc, _ = ops.cheaptrick_mcep(torch.from_numpy(x).to(gpu), [0, len(x)],
                                    torch.from_numpy(f0).to(gpu), [0, len(f0)], fs, order=59,
                                    alpha=0.554, mc_dtype=torch.float64)
    sp_ref = capi.cheaptrick(x, fs, tp, f0)
    assert sp.shape[1] == 1025
    assert np.abs(sp.cpu().numpy() / sp_ref - 1).max() < 1e-8
    mc_ref = capi.mcep(np.sqrt(sp_ref), 59, 0.554)
    assert np.sqrt(np.mean((mc.cpu().numpy() - mc_ref) ** 2)) < 1e-7


def test_dio_stonemask_match_oracle(gpu, utts):
    from idiaptts_amd import ops
    from oracle import capi
    x, _, x_off, f_off = _batch(utts, gpu)
    fs = utts[0][1]
    f0d = ops.dio(x, x_off, f_off, fs)
    f0r = ops.stonemask(x, x_off, f0d, f_off, fs).cpu().numpy()
    f0d = f0d.cpu().numpy()
    for u, (xu, _, f0_ref, tpu) in enumerate(utts):
        a, b = f_off[u], f_off[u + 1]
        d_ref, _ = capi.dio(xu, fs)
        assert np.array_equal(f0d[a:b] == 0, d_ref == 0)            # discrete decisions identical
        assert np.abs(f0d[a:b] - d_ref).max() < 1e-7
        assert np.array_equal(f0r[a:b] == 0, f0_ref == 0)
        assert np.abs(f0r[a:b] - f0_ref).max() < 1e-7


@pytest.mark.parametrize("fs_name", ["16k", "48k"])
def test_stonemask_below_dios_floor_takes_the_second_launch(gpu, utts, golden_dir, fs_name):
    """itts_stonemask sizes the LDS blocks of its main launch for f0 > 70 Hz (DIO's floor is 71 Hz) and gives the
    frames in (40 Hz, 70 Hz] -- which the interface admits and only a caller's own contour can hold -- to a second
    launch with the long blocks: a contour with such frames (and some outside (40 Hz, fs / 12]) against the oracle."""
    from idiaptts_amd import ops
    from oracle import capi
    if fs_name == "16k":
        xu, fs, _, tpu = utts[0]
    else:
        from scipy.io import wavfile
        fs, w = wavfile.read(os.path.join(golden_dir, "p225_001.wav"))      # the reference's own 48 kHz fixture
        assert fs == 48000
        xu = w.astype(np.float64) / 32768.0
    T = int(1000.0 * len(xu) / fs / 5.0) + 1
    tp = np.arange(T) * 0.005
    rng = np.random.default_rng(11)
    f0 = rng.choice([0.0, 30.0, 41.0, 47.5, 55.0, 69.9, 70.0, 70.1, 95.0, 180.0, fs / 12.0 + 1.0], size=T)
    ref = capi.stonemask(np.ascontiguousarray(xu), fs, tp, f0)
    got = ops.stonemask(torch.from_numpy(np.ascontiguousarray(xu)).to(gpu), [0, len(xu)],
                        torch.from_numpy(f0).to(gpu), [0, T], fs).cpu().numpy()
    assert np.array_equal(got == 0, ref == 0)
    assert np.abs(got - ref).max() < 1e-7
    assert (got[(f0 > 40) & (f0 <= 70)] > 0).any()        # the second launch did refine something


def test_d4c_matches_oracle(gpu, utts):
    from idiaptts_amd import ops
    from oracle import capi
    x, f0, x_off, f_off = _batch(utts, gpu)
    fs = utts[0][1]
    ap, bap = ops.d4c(x, x_off, f0, f_off, fs, want_bap=torch.float64)
    ap = ap.cpu().numpy()
    bap = bap.cpu().numpy()
    for u, (xu, _, f0u, tpu) in enumerate(utts):
        a, b = f_off[u], f_off[u + 1]
        ap_ref = capi.d4c(xu, fs, tpu, f0u)
        unv_ref = ap_ref[:, 0] > 0.999
        assert np.array_equal(ap[a:b, 0] > 0.999, unv_ref)           # LoveTrain V/UV identical
        assert np.abs(20 * np.log10(ap[a:b] / ap_ref)).max() < 1e-6  # dB
        bap_ref = capi.code_aperiodicity(ap_ref, fs)
        assert np.abs(bap[a:b] - bap_ref).max() < 1e-6


def test_full_analysis_chain_matches_reference_cmp(gpu, golden_dir):
    """wav -> DIO -> StoneMask -> CheapTrick -> mcep / D4C -> bap on the GPU, then the reference's
    host logic (lf0 threshold, interpolate_lin), against the golden .cmp of the reference."""
    import math
    from idiaptts_amd import ops
    from idiaptts_amd.misc.utils import interpolate_lin
    names = ["LJ001-000%d" % i for i in range(1, 10)]   # every fixture utterance of the reference
    xs = [_read(golden_dir, n)[0] for n in names]
    fs = 16000
    x_off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).tolist()
    T = [int(1000.0 * len(x) / fs / 5.0) + 1 for x in xs]
    f_off = np.concatenate([[0], np.cumsum(T)]).tolist()
    x = torch.from_numpy(np.concatenate(xs)).to(gpu)
    f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs), f_off, fs)
    _, mc, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, want_sp=False, order=19, alpha=0.58)
    _, bap = ops.d4c(x, x_off, f0, f_off, fs, want_ap=False, want_bap=torch.float32)
    f0 = f0.cpu().numpy()
    for u, name in enumerate(names):
        a, b = f_off[u], f_off[u + 1]
        cmp_ = np.fromfile(os.path.join(golden_dir, name + ".cmp"), dtype=np.float32).reshape(-1, 67)
        lf0 = np.log(f0[a:b].clip(min=1e-10), dtype=np.float32)
        lf0[lf0 <= math.log(30)] = 0
        lf0, vuv = interpolate_lin(lf0)
        assert np.array_equal(vuv[:, 0].astype(np.float32), cmp_[:, 63])        # bit-exact V/UV
        assert np.sqrt(np.mean((lf0[:, 0] - cmp_[:, 60]) ** 2)) < 1e-6          # bar 1e-4
        assert np.sqrt(np.mean((bap[a:b, 0].cpu().numpy() - cmp_[:, 64]) ** 2)) < 1e-6
        assert np.sqrt(np.mean((mc[a:b].cpu().numpy() - cmp_[:, :20]) ** 2)) < 1e-6
        assert np.abs(mc[a:b].cpu().numpy() - cmp_[:, :20]).max() <= 1e-6


def test_decode_aperiodicity_inverts_the_reference_held_bap(gpu, golden_dir):
    """itts_decode_aperiodicity on the bap column of every reference `.cmp`, coded again by
    itts_code_aperiodicity (the pinned direction): the round trip returns the stored values, and
    the HIP decoder equals the oracle's."""
    from idiaptts_amd import ops
    from oracle import capi
    fs = 16000
    for i in range(1, 10):
        cmp_ = np.fromfile(os.path.join(golden_dir, "LJ001-000%d.cmp" % i), dtype=np.float32).reshape(-1, 67)
        bap = cmp_[:, 64:65].astype(np.float64)
        ap = ops.decode_aperiodicity(torch.from_numpy(bap).to(gpu), fs, 1024)
        ap_ref = capi.decode_aperiodicity(bap, fs, 1024)
        assert np.abs(ap.cpu().numpy() - ap_ref).max() < 1e-12
        back = ops.code_aperiodicity(ap, fs).cpu().numpy()
        voiced = bap[:, 0] < -0.5
        assert np.abs(back[voiced] - bap[voiced]).max() < 1e-9
        assert np.abs(back[~voiced]).max() < 1e-9


def test_analysis_of_the_reference_48k_fixture_matches_oracle(gpu, golden_dir):
    """The reference's own 48 kHz audio (fixtures/database/wav48/p225_001.wav,
    test_WorldFeatLabelGen.py:611-629): the whole file, HIP against the C oracle -- identical V/UV
    decisions, f0 / bap / mcep within the north star's 1e-4 RMSE by orders of magnitude."""
    from idiaptts_amd import ops
    from oracle import capi
    fs, w = wavfile.read(os.path.join(golden_dir, "p225_001.wav"))
    assert fs == 48000
    x = w.astype(np.float64) / 32768.0
    T = int(1000.0 * len(x) / fs / 5.0) + 1
    xg = torch.from_numpy(x).to(gpu)
    f0 = ops.stonemask(xg, [0, len(x)], ops.dio(xg, [0, len(x)], [0, T], fs), [0, T], fs)
    f0_ref, sp_ref, ap_ref = capi.wav2world(x, fs)
    assert (f0_ref > 0).sum() > 100
    assert np.array_equal(f0.cpu().numpy() == 0, f0_ref == 0)
    assert np.abs(f0.cpu().numpy() - f0_ref).max() < 1e-6
    _, bap = ops.d4c(xg, [0, len(x)], f0, [0, T], fs, want_ap=False, want_bap=torch.float64)
    bap_ref = capi.code_aperiodicity(ap_ref, fs)
    assert bap.shape[1] == 5
    assert np.array_equal(bap.cpu().numpy()[:, 0] > -1e-6, bap_ref[:, 0] > -1e-6)   # LoveTrain V/UV
    assert np.sqrt(np.mean((bap.cpu().numpy() - bap_ref) ** 2)) < 1e-5
    sp, mc, iters = ops.cheaptrick_mcep(xg, [0, len(x)], f0, [0, T], fs, order=59, alpha=0.77,
                                        mc_dtype=torch.float64, want_iters=True)
    assert np.abs(sp.cpu().numpy() / sp_ref - 1).max() < 1e-6
    mc_ref, it_ref = capi.mcep(np.sqrt(sp_ref), 59, 0.77, return_iters=True)
    assert np.array_equal(iters.cpu().numpy(), it_ref)                              # same Newton trip counts
    assert np.sqrt(np.mean((mc.cpu().numpy() - mc_ref) ** 2)) < 1e-7


def test_analysis_48k_synthetic_matches_oracle(gpu):
    from idiaptts_amd import ops
    from oracle import capi
    fs = 48000
    x = _synthetic(fs, 1.0, 2)
    T = int(1000.0 * len(x) / fs / 5.0) + 1
    xg = torch.from_numpy(x).to(gpu)
    f0 = ops.stonemask(xg, [0, len(x)], ops.dio(xg, [0, len(x)], [0, T], fs), [0, T], fs)
    f0_ref, sp_ref, ap_ref = capi.wav2world(x, fs)
    assert np.array_equal(f0.cpu().numpy() == 0, f0_ref == 0)
    assert np.abs(f0.cpu().numpy() - f0_ref).max() < 1e-6
    _, bap = ops.d4c(xg, [0, len(x)], f0, [0, T], fs, want_ap=False, want_bap=torch.float64)
    bap_ref = capi.code_aperiodicity(ap_ref, fs)
    assert bap.shape[1] == 5
    assert np.sqrt(np.mean((bap.cpu().numpy() - bap_ref) ** 2)) < 1e-5


def test_synthesis_matches_oracle(gpu, golden_dir):
    """No golden waveform exists in the reference (parity unpinned there); the HIP synthesis must
    reproduce the C oracle's samples: same pulse positions, same xorshift stream (jump-ahead),
    RMSE far below the 1e-4 north-star bar."""
    from idiaptts_amd import ops
    from oracle import capi
    ys, f0s, sps, aps = [], [], [], []
    for name in ["LJ001-0008", "LJ001-0002"]:
        fs, w = wavfile.read(os.path.join(golden_dir, name + ".wav"))
        raw = w.astype(np.float64) / 32768.0
        f0, sp, ap = capi.wav2world(raw, fs)
        ys.append(capi.synthesize(f0, sp, ap, fs))
        f0s.append(f0); sps.append(sp); aps.append(ap)
    f_off = np.concatenate([[0], np.cumsum([len(f) for f in f0s])]).tolist()
    y, y_off = ops.world_synthesize(torch.from_numpy(np.concatenate(f0s)).to(gpu),
                                    torch.from_numpy(np.concatenate(sps)).to(gpu),
                                    torch.from_numpy(np.concatenate(aps)).to(gpu), f_off, 16000,
                                    dtype=torch.float64)
    y = y.cpu().numpy()
    for u in range(2):
        a, b = y_off[u], y_off[u + 1]
        assert b - a == len(ys[u])
        ref = ys[u].astype(np.float32).astype(np.float64)
        rmse = np.sqrt(np.mean((y[a:b] - ref) ** 2))
        assert rmse < 1e-7, rmse
        assert np.abs(y[a:b] - ref).max() < 1e-6
    # de-pre-emphasis path == scipy.signal.lfilter on the float32 samples
    import scipy.signal
    y2, _ = ops.world_synthesize(torch.from_numpy(f0s[0]).to(gpu), torch.from_numpy(sps[0]).to(gpu),
                                 torch.from_numpy(aps[0]).to(gpu), [0, len(f0s[0])], 16000,
                                 preemphasis=0.97, dtype=torch.float64)
    ref2 = scipy.signal.lfilter([1], [1, -0.97], ys[0].astype(np.float32))
    assert np.abs(y2.cpu().numpy() - ref2).max() < 1e-5
    # the filter runs segment-parallel with a warm-up (0.97^warm <= 2^-64): against the strictly
    # sequential recurrence on the kernel's own f32-rounded samples the result is exact
    y0, _ = ops.world_synthesize(torch.from_numpy(f0s[0]).to(gpu), torch.from_numpy(sps[0]).to(gpu),
                                 torch.from_numpy(aps[0]).to(gpu), [0, len(f0s[0])], 16000,
                                 dtype=torch.float64)
    seq = np.empty(len(y0))
    prev = 0.0
    for i, v in enumerate(y0.cpu().numpy()):
        prev = v + 0.97 * prev
        seq[i] = prev
    assert np.array_equal(y2.cpu().numpy(), seq)
    for pre in (0.5, 0.999, -0.9, 1.0):      # short / very long memory (single segment), odd cases
        yp, _ = ops.world_synthesize(torch.from_numpy(f0s[0]).to(gpu), torch.from_numpy(sps[0]).to(gpu),
                                     torch.from_numpy(aps[0]).to(gpu), [0, len(f0s[0])], 16000,
                                     preemphasis=pre, dtype=torch.float64)
        refp = scipy.signal.lfilter([1], [1, -pre], y0.cpu().numpy())
        assert np.abs(yp.cpu().numpy() - refp).max() < 1e-9 * max(1.0, np.abs(refp).max())


def test_synthesis_48k(gpu):
    from idiaptts_amd import ops
    from oracle import capi
    fs = 48000
    x = _synthetic(fs, 0.6, 3)
    f0, sp, ap = capi.wav2world(x, fs)
    ref = capi.synthesize(f0, sp, ap, fs)
    y, y_off = ops.world_synthesize(torch.from_numpy(f0).to(gpu), torch.from_numpy(sp).to(gpu),
                                    torch.from_numpy(ap).to(gpu), [0, len(f0)], fs,
                                    dtype=torch.float64)
    assert y_off[-1] == len(ref)
    assert np.sqrt(np.mean((y.cpu().numpy() - ref.astype(np.float32)) ** 2)) < 1e-7


def test_parallel_phase_scan_is_bit_identical_to_the_sequential_chain(gpu):
    """WORLD's pulse positions hang on the sequential rounding of the running phase sum
    (synthesis.cpp GetTemporalParametersForTimeBase); the scan kernel reproduces that chain exactly
    in integer units of the current ulp (ties, binade crossings).  Whole waveforms from the scan
    and from the strictly sequential kernel (ITTS_SYNTH_SEQ_PHASE) must be the same bits, on
    contours that exercise long unvoiced runs (constant 500 Hz: the sum hits multiples of 2 pi),
    random voiced contours and very short utterances."""
    from idiaptts_amd import ops
    rng = np.random.default_rng(11)
    fs, K = 16000, 513
    f0s = []
    for T in (2, 3, 7, 400, 1500, 2000):
        f0 = np.clip(150 + np.cumsum(rng.normal(0, 3, T)), 60, 400)
        unv = np.zeros(T, dtype=bool)
        p = 0
        while p < T:
            seg = int(rng.integers(5, 120))
            if rng.uniform() < 0.45:
                unv[p:p + seg] = True
            p += seg
        f0[unv] = 0.0
        f0s.append(f0)
    f0s.append(np.zeros(900))                       # all unvoiced
    f0s.append(np.full(900, 71.0))                  # lowest F0, all voiced
    f_off = np.concatenate([[0], np.cumsum([len(f) for f in f0s])]).tolist()
    n = f_off[-1]
    sp = torch.from_numpy(np.abs(rng.normal(1e-3, 2e-4, size=(n, K))) + 1e-5).to(gpu)
    ap = torch.from_numpy(rng.uniform(0.01, 0.95, size=(n, K))).to(gpu)
    f0 = torch.from_numpy(np.concatenate(f0s)).to(gpu)
    os.environ.pop("ITTS_SYNTH_SEQ_PHASE", None)
    y_scan, y_off = ops.world_synthesize(f0, sp, ap, f_off, fs, dtype=torch.float64)
    os.environ["ITTS_SYNTH_SEQ_PHASE"] = "1"
    try:
        y_seq, _ = ops.world_synthesize(f0, sp, ap, f_off, fs, dtype=torch.float64)
    finally:
        os.environ.pop("ITTS_SYNTH_SEQ_PHASE", None)
    assert torch.isfinite(y_scan).all() and y_scan.abs().max() > 0
    # identical pulse positions and phases -> identical waveforms up to the order in which the f64
    # atomics of overlapping pulses land (a pulse moved by one sample would show as ~1e-3)
    assert (y_scan - y_seq).abs().max() < 1e-13


def test_ragged_batch_with_silence_and_very_short_utterances(gpu):
    """Edge cases of the batched entry points: a digitally silent utterance (all frames unvoiced,
    CheapTrick on its default F0, D4C's LoveTrain rejects every frame), utterances of a handful of
    frames (DIO's decimated signal shorter than its filters), next to a normal one.  Every
    utterance must equal the oracle run on it alone (no leakage across the batch), and an empty
    batch is a no-op."""
    from idiaptts_amd import ops
    from oracle import capi
    fs = 16000
    xs = [_synthetic(fs, 0.5, 5), np.zeros(4000), _synthetic(fs, 0.05, 6), _synthetic(fs, 0.011, 7),
          1e-3 * np.random.default_rng(3).normal(size=2400)]
    T = [int(1000.0 * len(x) / fs / 5.0) + 1 for x in xs]
    x_off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).tolist()
    f_off = np.concatenate([[0], np.cumsum(T)]).tolist()
    x = torch.from_numpy(np.concatenate(xs)).to(gpu)
    f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs), f_off, fs)
    sp, mc, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, order=19, alpha=0.58,
                                    mc_dtype=torch.float64)
    ap, bap = ops.d4c(x, x_off, f0, f_off, fs, want_bap=torch.float64)
    y, y_off = ops.world_synthesize(f0, sp, ap, f_off, fs, dtype=torch.float64)
    f0, sp, mc, ap, bap, y = (t.cpu().numpy() for t in (f0, sp, mc, ap, bap, y))
    for u, xu in enumerate(xs):
        a, b = f_off[u], f_off[u + 1]
        f0_ref, sp_ref, ap_ref = capi.wav2world(xu, fs)
        assert len(f0_ref) == b - a
        assert np.array_equal(f0[a:b] == 0, f0_ref == 0), u
        assert np.abs(f0[a:b] - f0_ref).max() < 1e-7, u
        assert np.abs(np.log(sp[a:b] / sp_ref)).max() < 1e-8, u
        assert np.abs(20 * np.log10(ap[a:b] / ap_ref)).max() < 1e-6, u
        assert np.abs(bap[a:b] - capi.code_aperiodicity(ap_ref, fs)).max() < 1e-6, u
        assert np.abs(mc[a:b] - capi.mcep(np.sqrt(sp_ref), 19, 0.58)).max() < 1e-6, u
        y_ref = capi.synthesize(f0_ref, sp_ref, ap_ref, fs).astype(np.float32)
        assert y_off[u + 1] - y_off[u] == len(y_ref)
        assert np.abs(y[y_off[u]:y_off[u + 1]] - y_ref).max() < 1e-6, u
    assert (f0[f_off[1]:f_off[2]] == 0).all()                      # silence: nothing voiced
    # empty batch
    e = torch.empty((0,), dtype=torch.float64, device=gpu)
    assert ops.dio(e, [0], [0], fs).numel() == 0
    assert ops.stonemask(e, [0], e, [0], fs).numel() == 0
    sp0, mc0, _ = ops.cheaptrick_mcep(e, [0], e, [0], fs, order=19, alpha=0.58)
    assert sp0.shape == (0, 513) and mc0.shape == (0, 20)
    ap0, _ = ops.d4c(e, [0], e, [0], fs)
    assert ap0.shape == (0, 513)
    y0, y0_off = ops.world_synthesize(e, sp0, ap0, [0], fs)
    assert y0.numel() == 0 and y0_off == [0]


def test_cheaptrick_frames_with_f0_beyond_nyquist_take_the_workgroup_kernel(gpu, utts):
    """F0 at or above fs / 2 is an input error (WORLD's DC correction reads past its arrays there, so the
    oracle has no answer to compare with); the wave-per-frame kernel leaves such frames to the
    workgroup kernel (csrc/world_frame.hip: ct_far).  The call must go through, and every other frame
    must come out bit for bit as it does without the stray value -- the last frame of an utterance is
    changed, so that no other frame's position in the safeguard-noise stream moves."""
    from idiaptts_amd import ops
    x, f0, x_off, f_off = _batch(utts, gpu)
    fs = utts[0][1]
    sp_a, _, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs)
    f0_b = f0.clone()
    last = f_off[1] - 1
    f0_b[last] = 0.6 * fs
    sp_b, _, _ = ops.cheaptrick_mcep(x, x_off, f0_b, f_off, fs)
    keep = torch.ones(f0.numel(), dtype=torch.bool, device=gpu)
    keep[last] = False
    assert torch.equal(sp_a[keep], sp_b[keep])
    assert not torch.equal(sp_a[last], sp_b[last])


def test_long_utterances_exercise_the_large_size_paths(gpu):
    """Sizes past the small-case shortcuts: (1) a 9.5-minute digitally silent utterance -- its
    CheapTrick envelope is nothing but WORLD's safeguard noise, 68 M normals into the stream, i.e.
    past the tabulated generator states (2^20 chunks), so the GF(2) jump matrices decide the last
    frames; (2) a 50 s utterance (10 k frames) through DIO, whose contour kernel then keeps its
    candidates and boundary masks in global scratch instead of LDS."""
    from idiaptts_amd import ops
    from oracle import capi
    fs = 16000
    n = int(fs * 570)
    T = int(1000.0 * n / fs / 5.0) + 1
    x = torch.zeros(n, dtype=torch.float64, device=gpu)
    f0 = torch.zeros(T, dtype=torch.float64, device=gpu)
    sp, _, _ = ops.cheaptrick_mcep(x, [0, n], f0, [0, T], fs)
    assert T * (97 + 513) > 64 * 2 ** 20                       # beyond the state table
    tp = np.arange(T) * 0.005
    sp_ref = capi.cheaptrick(np.zeros(n), fs, tp, np.zeros(T))
    got = sp.cpu().numpy()
    for sl in (slice(0, 50), slice(T // 2, T // 2 + 50), slice(T - 50, T)):
        assert np.abs(np.log(got[sl] / sp_ref[sl])).max() < 1e-9
    del sp, got, sp_ref
    xl = _synthetic(fs, 50.0, 9)
    Tl = int(1000.0 * len(xl) / fs / 5.0) + 1
    assert Tl > 8192
    xg = torch.from_numpy(xl).to(gpu)
    f0d = ops.dio(xg, [0, len(xl)], [0, Tl], fs).cpu().numpy()
    d_ref, _ = capi.dio(xl, fs)
    assert np.array_equal(f0d == 0, d_ref == 0) and np.abs(f0d - d_ref).max() < 1e-7


@pytest.mark.parametrize("fs,nap", [(22050, 2), (24000, 3), (44100, 5)])
def test_other_sampling_rates_match_oracle(gpu, fs, nap):
    """22.05 / 24 / 44.1 kHz (the reference's fs_to_num_bap / fs_to_frame_length tables,
    AudioProcessing.py:52-71): 2 / 3 / 5 aperiodicity bands, 1024- / 2048-point envelopes."""
    from idiaptts_amd import ops
    from oracle import capi
    x = _synthetic(fs, 0.6, 4)
    T = int(1000.0 * len(x) / fs / 5.0) + 1
    xg = torch.from_numpy(x).to(gpu)
    f0 = ops.stonemask(xg, [0, len(x)], ops.dio(xg, [0, len(x)], [0, T], fs), [0, T], fs)
    f0_ref, sp_ref, ap_ref = capi.wav2world(x, fs)
    assert np.array_equal(f0.cpu().numpy() == 0, f0_ref == 0)
    assert np.abs(f0.cpu().numpy() - f0_ref).max() < 1e-6
    sp, mc, _ = ops.cheaptrick_mcep(xg, [0, len(x)], f0, [0, T], fs, order=59, alpha=0.5,
                                    mc_dtype=torch.float64)
    assert sp.shape[1] == sp_ref.shape[1]
    assert np.abs(np.log(sp.cpu().numpy() / sp_ref)).max() < 1e-7
    assert np.abs(mc.cpu().numpy() - capi.mcep(np.sqrt(sp_ref), 59, 0.5)).max() < 1e-6
    ap, bap = ops.d4c(xg, [0, len(x)], f0, [0, T], fs, want_bap=torch.float64)
    assert bap.shape[1] == nap
    assert np.abs(bap.cpu().numpy() - capi.code_aperiodicity(ap_ref, fs)).max() < 1e-5
    y, _ = ops.world_synthesize(torch.from_numpy(f0_ref).to(gpu), torch.from_numpy(sp_ref).to(gpu),
                                torch.from_numpy(ap_ref).to(gpu), [0, T], fs, dtype=torch.float64)
    y_ref = capi.synthesize(f0_ref, sp_ref, ap_ref, fs).astype(np.float32)
    assert np.abs(y.cpu().numpy() - y_ref).max() < 1e-6


def test_wav2world_composite_equals_the_separate_calls_and_the_oracle(gpu, golden_dir):
    """itts_wav2world (pyworld.wav2world, WorldFeatLabelGen.py:792-793) is DIO -> StoneMask ->
    CheapTrick -> D4C in one call: identical to the separate entry points, and equal to the C
    oracle's wav2world on a fixture clip."""
    from idiaptts_amd import ops
    from oracle import capi
    xs = [_read(golden_dir, "LJ001-0008")[0][:24000], _read(golden_dir, "LJ001-0002")[0][8000:20000]]
    fs = 16000
    x_off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).tolist()
    f_off = np.concatenate([[0], np.cumsum([int(1000.0 * len(x) / fs / 5.0) + 1 for x in xs])]).tolist()
    x = torch.from_numpy(np.concatenate(xs)).to(gpu)
    f0, sp, ap = ops.wav2world(x, x_off, f_off, fs)
    f0_s = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs), f_off, fs)
    sp_s, _, _ = ops.cheaptrick_mcep(x, x_off, f0_s, f_off, fs, want_sp=True)
    ap_s, _ = ops.d4c(x, x_off, f0_s, f_off, fs, want_ap=True)
    assert torch.equal(f0, f0_s) and torch.equal(sp, sp_s) and torch.equal(ap, ap_s)
    f0_only, none_sp, ap_only = ops.wav2world(x, x_off, f_off, fs, want_sp=False)
    assert none_sp is None and torch.equal(f0_only, f0) and torch.equal(ap_only, ap)
    for u, xu in enumerate(xs):
        a, b = f_off[u], f_off[u + 1]
        f0_o, sp_o, ap_o = capi.wav2world(xu, fs)
        assert np.array_equal(f0[a:b].cpu().numpy() > 0, f0_o > 0)
        assert np.abs(f0[a:b].cpu().numpy() - f0_o).max() < 1e-6
        assert np.abs(sp[a:b].cpu().numpy() / sp_o - 1).max() < 1e-6
        assert np.abs(ap[a:b].cpu().numpy() - ap_o).max() < 1e-6
