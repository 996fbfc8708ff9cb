"""The premise behind FreqtTables::specT (csrc/context.hip): the log amplitude
pysptk.mgc2sp(mc, alpha, 0, fftlen).real (AudioProcessing.py:252-256) -- de-warping freqt(mc, -alpha), transform,
real part -- is a LINEAR map of the mel-cepstrum, so the matrix of its values on the unit vectors reproduces
it for any mc as one product.  Checked on the CPU against the C oracle's mgc2sp (the transform version);
the device's matrix is built from the same definition and is covered by tests/test_gpu_world.py."""
import numpy as np
import pytest


@pytest.mark.parametrize("order,alpha,fftlen", [(59, 0.58, 1024), (19, 0.41, 512), (24, 0.77, 2048)])
def test_mgc2sp_log_amplitude_is_linear_in_the_mel_cepstrum(order, alpha, fftlen):
    from oracle import capi
    m1 = order + 1
    spec_t = capi.mgc2sp_logamp(np.eye(m1), alpha, fftlen)            # row j: the map's value on e_j
    assert spec_t.shape == (m1, fftlen // 2 + 1)
    rng = np.random.default_rng(order)
    mc = rng.normal(size=(40, m1)) * np.exp(-np.arange(m1) / 12.0)     # decaying like real mel-cepstra
    ref = capi.mgc2sp_logamp(mc, alpha, fftlen)
    got = mc @ spec_t
    assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    # the zeroth row is the constant 1 (c0 passes through the all-pass warping unchanged)
    assert np.abs(spec_t[0] - 1.0).max() < 1e-12
