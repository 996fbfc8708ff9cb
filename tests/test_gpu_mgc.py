"""Mel-generalized cepstral analysis / synthesis filters spectra on the GPU (csrc/mgcep.hip)
against the C oracle (oracle/c/sptk.c: orc_mgcep, orc_mgc2sp_gamma) and the bound the reference's
own test applies (test_WorldFeatLabelGen.py:827-836).  PARITY UNPINNED for gamma != 0 (pysptk is
not installable, the reference holds no MGC vector); see tests/test_oracle_golden.py for what pins
the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _envelopes(golden_dir, n_frames=40, order=19):
    from oracle import capi
    cmp_ = np.fromfile(os.path.join(golden_dir, "LJ001-0008.cmp"), dtype=np.float32).reshape(-1, 67)
    mc = cmp_[100:100 + n_frames, :20].astype(np.float64)
    return np.exp(capi.mgc2sp_logamp(mc, 0.58, 1024))


@pytest.mark.parametrize("gamma,order,alpha", [(-1.0 / 3.0, 19, 0.58), (-1.0 / 3.0, 59, 0.58),
                                               (-0.5, 24, 0.42), (0.0, 19, 0.58), (-1.0, 19, 0.58),
                                               (-1.0 / 3.0, 19, 0.0)])
def test_mgcep_matches_oracle(gpu, golden_dir, gamma, order, alpha):
    from idiaptts_amd import ops
    from oracle import capi
    amp = _envelopes(golden_dir)
    want, it_ref = capi.mgcep(amp, order, alpha, gamma, return_iters=True)
    got, it = ops.mgcep(torch.from_numpy(amp).to(gpu), order, alpha, gamma, dtype=torch.float64,
                        want_iters=True)
    assert np.array_equal(it.cpu().numpy(), it_ref)                   # same Newton trip counts
    assert np.abs(got.cpu().numpy() - want).max() < 1e-8 * max(1.0, np.abs(want).max())
    got32 = ops.mgcep(torch.from_numpy(amp ** 2).to(gpu), order, alpha, gamma, input_is_power=True)
    assert got32.dtype == torch.float32
    assert np.abs(got32.cpu().numpy() - want.astype(np.float32)).max() < 1e-5


@pytest.mark.parametrize("gamma,alpha,fftlen", [(-1.0 / 3.0, 0.58, 1024), (-0.5, 0.42, 512),
                                                (-1.0 / 3.0, 0.0, 1024), (0.0, 0.58, 1024),
                                                (-1.0, 0.3, 1024)])
def test_mgc2sp_gamma_matches_oracle(gpu, golden_dir, gamma, alpha, fftlen):
    from idiaptts_amd import ops
    from oracle import capi
    amp = _envelopes(golden_dir, 12)
    mgc = capi.mgcep(amp, 19, alpha, gamma)
    want = capi.mgc2sp_gamma_logamp(mgc, alpha, gamma, fftlen)
    got = ops.mgc2sp_gamma(torch.from_numpy(mgc).to(gpu), alpha, gamma, fftlen, want_logamp=True)
    assert np.abs(got.cpu().numpy() - want).max() < 1e-9 * max(1.0, np.abs(want).max())
    a32 = ops.mgc2sp_gamma(torch.from_numpy(mgc).to(gpu), alpha, gamma, fftlen).cpu().numpy()
    assert np.allclose(a32, np.exp(want.astype(np.float32)), rtol=2e-6)


def test_extract_mgc_and_mgc_to_amp_sp_reference_bound(gpu, golden_dir):
    """AudioProcessing.extract_mgc / mgc_to_amp_sp / decode_sp(sp_type='mgc') on the fixture wav:
    the reconstruction stays inside the bound of the reference's test (sum of squared amplitude
    errors < 1500 for the utterance, test_WorldFeatLabelGen.py:827-836; there against a librosa
    spectrum, here against the WORLD envelope the coefficients were extracted from)."""
    from scipy.io import wavfile
    from idiaptts_amd.src.data_preparation.audio.AudioProcessing import AudioProcessing
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    fs, w = wavfile.read(os.path.join(golden_dir, "LJ001-0008.wav"))
    raw = w.astype(np.float64) / 32768.0
    amp_sp, lf0, vuv, bap = WorldFeatLabelGen.world_extract_features(raw, fs, 5)
    mgc = AudioProcessing.extract_mgc(amp_sp, fs=fs, num_coded_sps=60)
    assert mgc.dtype == np.float32 and mgc.shape == (len(amp_sp), 60)
    rec = AudioProcessing.mgc_to_amp_sp(mgc, fs)
    assert rec.dtype == np.float32 and rec.shape == amp_sp.shape
    assert ((amp_sp - rec) ** 2).sum() < 1500
    assert np.array_equal(AudioProcessing.decode_sp(mgc, "mgc", fs), rec)
    # the batched feature path with sp_type 'mgc' produces the same coefficients
    feats = WorldFeatLabelGen.extract_features_batch([raw], fs, sp_type="mgc", num_coded_sps=60)[0]
    assert np.abs(feats[0] - mgc).max() < 1e-5
