"""Synthetic speech-like audio of the bench workloads; numpy / scipy only, so
that CPU baseline workers can import it without loading torch."""
import numpy as np


# --------------------------------------------------------------------------- synthetic audio
def make_audio(fs, seconds, seed):
    """SURVEY.md section 8d signal: harmonic source with an F0 random walk in [90, 300] Hz, ~60 %
    voiced in 0.2-1 s segments, shaped by a random stable AR envelope, plus -40 dB white noise,
    amplitude 0.3."""
    import scipy.signal
    rng = np.random.default_rng(1234 + seed)
    n = int(fs * seconds)
    f0 = np.clip(180.0 + np.cumsum(rng.normal(0.0, 0.03, n)) * 25.0, 90.0, 300.0)
    voiced = np.zeros(n)
    pos = 0
    while pos < n:
        seg = int(fs * rng.uniform(0.2, 1.0))
        voiced[pos:pos + seg] = 1.0 if rng.uniform() < 0.6 else 0.0
        pos += seg
    phase = 2.0 * np.pi * np.cumsum(f0) / fs
    src = np.zeros(n)
    for k in range(1, 16):
        src += np.sin(k * phase) / k
    src *= voiced
    # random stable 20-pole AR envelope: 10 conjugate pole pairs inside the unit circle
    poles = rng.uniform(0.80, 0.97, 10) * np.exp(1j * rng.uniform(0.05, 0.95, 10) * np.pi)
    a = np.real(np.poly(np.concatenate([poles, np.conj(poles)])))
    y = scipy.signal.lfilter([1.0], a, src + 0.05 * rng.normal(size=n))
    y = 0.3 * y / (np.abs(y).max() + 1e-12)
    return y + 10.0 ** (-40.0 / 20.0) * rng.normal(size=n)


def make_audio_batch(n_utts, fs, seed=0, min_s=2.0, max_s=10.0):
    rng = np.random.default_rng(99 + seed)
    durs = rng.uniform(min_s, max_s, size=n_utts)
    return [make_audio(fs, float(d), seed * 100000 + i) for i, d in enumerate(durs)]
