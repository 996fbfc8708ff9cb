"""AudioProcessing with the reference's interface for the WORLD hot path
(idiaptts/src/data_preparation/audio/AudioProcessing.py); pysptk / pyworld calls are replaced
by the HIP kernels behind libidiaptts_amd.so. librosa-based helpers (mel filter banks,
Griffin-Lim) are out of scope."""
import logging
import os

import numpy as np
import scipy.io.wavfile
import scipy.signal
import torch

from .... import lib as _lib
from .... import ops


def _dev():
    _lib.require_gpu()
    return torch.device("cuda")


def merlin_post_filter(mgc, alpha, minimum_phase_order=511, fftlen=1024, coef=1.4, weight=None):
    """Merlin's formant-enhancing post filter on mel-cepstra [T, D] (nnmnkwii.postfilters.
    merlin_post_filter, called by the reference's decode_sp, AudioProcessing.py:308-314; nnmnkwii is
    not vendored: parity unpinned, the algorithm is restated from its published form):
    coefficients >= 2 are scaled by `coef`, then the 0th MLSA coefficient is corrected so that the
    frame keeps its energy r0 = c2acr(freqt(mc, -alpha))[0].

    The two energies come from the mgc2sp kernel (power spectrum of the de-warped cepstrum; its
    mean over the fftlen bins is the 0th autocorrelation); mc2b / b2mc are O(D) recursions over
    the coefficient index, vectorised over the frames on the host.  The de-warped cepstrum is
    truncated at fftlen/2 instead of minimum_phase_order: the omitted coefficient is ~ alpha^450."""
    mgc = np.ascontiguousarray(mgc, dtype=np.float64)
    T, D = mgc.shape
    if weight is None:
        weight = np.ones(D) * coef
        weight[:2] = 1
    assert len(weight) == D
    dev = _dev()

    def r0(mc):
        p = torch.exp(2.0 * ops.mgc2sp(torch.from_numpy(np.ascontiguousarray(mc)).to(dev), alpha,
                                       fftlen, want_logamp=True))
        return ((p[:, 0] + p[:, -1] + 2.0 * p[:, 1:-1].sum(dim=1)) / fftlen).cpu().numpy()

    def mc2b(mc):
        b = mc.copy()
        for m in range(D - 2, -1, -1):
            b[:, m] = mc[:, m] - alpha * b[:, m + 1]
        return b

    def b2mc(b):
        mc = b.copy()
        mc[:, :-1] = b[:, :-1] + alpha * b[:, 1:]
        return mc

    weighted = mgc * weight
    b = mc2b(weighted)
    b[:, 0] = np.log(r0(mgc) / r0(weighted)) / 2 + b[:, 0]
    return b2mc(b)


class AudioProcessing:
    mgc_gamma = -1. / 3.

    _alpha_cache = {}

    @staticmethod
    def fs_to_mgc_alpha(fs):
        """pysptk.util.mcepalpha(fs) (reference :32-40); a 23 ms search, cached per rate."""
        fs = int(fs)
        if fs not in AudioProcessing._alpha_cache:
            AudioProcessing._alpha_cache[fs] = _lib.load().itts_mcep_alpha(fs)
        return AudioProcessing._alpha_cache[fs]

    @staticmethod
    def fs_to_frame_length(fs):
        """pyworld.get_cheaptrick_fft_size(fs) (reference :52-60)."""
        return _lib.load().itts_cheaptrick_fft_size(int(fs), 71.0)

    @staticmethod
    def fs_to_num_bap(fs: int):
        """pyworld.get_num_aperiodicities(fs) (reference :69-71)."""
        return _lib.load().itts_num_aperiodicities(int(fs))

    @staticmethod
    def read_wav(audio_name):
        """soundfile.read equivalent for PCM / float wav files: samples in [-1, 1] as float64."""
        fs, data = scipy.io.wavfile.read(audio_name)
        if data.dtype == np.int16:
            raw = data.astype(np.float64) / 32768.0
        elif data.dtype == np.int32:
            raw = data.astype(np.float64) / 2147483648.0
        elif data.dtype == np.uint8:
            raw = (data.astype(np.float64) - 128.0) / 128.0
        else:
            raw = data.astype(np.float64)
        return raw, fs

    @staticmethod
    def get_raw(audio_name: str, preemphasis: float = 0.0):
        """Raw audio in [-1, 1] with pre-emphasis (reference :107-120)."""
        raw, fs = AudioProcessing.read_wav(audio_name)
        raw = np.append(raw[0], raw[1:] - preemphasis * raw[:-1])
        return raw, fs

    @staticmethod
    def get_raw_batch(audio_names, preemphasis: float = 0.0, n_threads: int = 8, pinned: bool = False):
        """get_raw for a list of files in one native call (csrc/hostio.cpp, a pool of plain
        threads): returns (samples of all files back to back as one float64 array, sample offsets
        [n+1], sampling rates).  Files the native reader does not take (multi-channel, exotic
        encodings) are read by get_raw.  pinned: the samples come as a page-locked torch tensor
        (float64) the device can fetch asynchronously (gen_data's reader threads; a pageable 51-MB
        batch took 4-6 ms of every batch to upload)."""
        import ctypes
        L = _lib.load()
        n = len(audio_names)
        fs = ctypes.c_int()
        ns = ctypes.c_int64()
        lengths, rates = [], []
        fallback = {}
        for i, name in enumerate(audio_names):
            rc = L.itts_wav_info(os.fsencode(name), ctypes.byref(fs), ctypes.byref(ns))
            if rc == 0:
                lengths.append(ns.value)
                rates.append(fs.value)
            elif rc == -3:                      # ITTS_E_UNSUPPORTED
                raw, rate = AudioProcessing.get_raw(name, preemphasis)
                fallback[i] = raw
                lengths.append(len(raw))
                rates.append(rate)
            else:
                _lib.check(rc, "itts_wav_info")
        offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        pinned_buf = None
        if pinned and torch.cuda.is_available():
            pinned_buf = torch.empty(int(offsets[-1]), dtype=torch.float64, pin_memory=True)
            out = pinned_buf.numpy()
        else:
            out = np.empty(int(offsets[-1]), dtype=np.float64)
        for i, raw in fallback.items():
            out[offsets[i]:offsets[i + 1]] = raw
        # the native call fills consecutive files: one call per run of natively readable files
        run_start = None
        for i in list(range(n)) + [n]:
            is_native = i < n and i not in fallback
            if is_native and run_start is None:
                run_start = i
            elif not is_native and run_start is not None:
                a, b = run_start, i
                sub = (ctypes.c_char_p * (b - a))(*[os.fsencode(audio_names[k])
                                                    for k in range(a, b)])
                offs = (ctypes.c_int64 * (b - a + 1))(*[int(o) for o in offsets[a:b + 1]])
                _lib.check(L.itts_wav_read_batch(sub, b - a, offs, float(preemphasis),
                                                 out.ctypes.data, int(n_threads)),
                           "itts_wav_read_batch")
                run_start = None
        return (pinned_buf if pinned_buf is not None else out), offsets, rates

    @staticmethod
    def extract_mcep(amp_sp: np.array, num_coded_sps: int, mgc_alpha: float) -> np.array:
        """pysptk.mcep(amp_sp, order, alpha, eps=1e-8, min_det=0, etype=1, itype=3) as float32
        (reference :142-153)."""
        a = torch.from_numpy(np.ascontiguousarray(amp_sp, dtype=np.float64)).to(_dev())
        return ops.mcep(a, num_coded_sps - 1, mgc_alpha, eps=1.0e-8).cpu().numpy()

    @staticmethod
    def extract_mgc(amp_sp: np.array, fs: int = None, num_coded_sps: int = 60,
                    mgc_alpha: float = None) -> np.array:
        """pysptk.mgcep(amp_sp, order, alpha, gamma=-1/3, eps=1e-8, min_det=0, etype=1, itype=3)
        as float32 (reference :123-140)."""
        if mgc_alpha is None:
            assert fs is not None, "Either sampling rate or mgc alpha has to be given."
            mgc_alpha = AudioProcessing.fs_to_mgc_alpha(fs)
        a = torch.from_numpy(np.ascontiguousarray(amp_sp, dtype=np.float64)).to(_dev())
        return ops.mgcep(a, num_coded_sps - 1, mgc_alpha, AudioProcessing.mgc_gamma,
                         eps=1.0e-8).cpu().numpy()

    @staticmethod
    def mgc_to_amp_sp(mgc: np.array, fs: int, alpha: float = None, gamma: float = None,
                      n_fft: int = None):
        """exp(float32(pysptk.mgc2sp(mgc, alpha, gamma, fftlen).real)) (reference :259-275)."""
        if alpha is None:
            alpha = AudioProcessing.fs_to_mgc_alpha(fs)
        if gamma is None:
            gamma = AudioProcessing.mgc_gamma
        if n_fft is None:
            n_fft = AudioProcessing.fs_to_frame_length(fs)
        m = torch.from_numpy(np.ascontiguousarray(mgc, dtype=np.float64)).to(_dev())
        return ops.mgc2sp_gamma(m, alpha, gamma, n_fft).cpu().numpy()

    @staticmethod
    def mcep_to_amp_sp(mcep: np.array, fs: int, alpha: float = None):
        """exp(float32(pysptk.mgc2sp(mcep, alpha, 0, fftlen).real)) (reference :247-256)."""
        if alpha is None:
            alpha = AudioProcessing.fs_to_mgc_alpha(fs)
        m = torch.from_numpy(np.ascontiguousarray(mcep, dtype=np.float64)).to(_dev())
        return ops.mgc2sp(m, alpha, AudioProcessing.fs_to_frame_length(fs)).cpu().numpy()

    @staticmethod
    def decode_sp(coded_sp: np.array, sp_type: str = "mcep", fs: int = None, alpha: float = None,
                  mgc_gamma: float = None, n_fft: int = None, post_filtering: bool = False):
        """reference :303-327"""
        if post_filtering:
            if sp_type in ("mcep", "mgc"):
                coded_sp = merlin_post_filter(coded_sp, AudioProcessing.fs_to_mgc_alpha(fs))
            else:
                logging.warning("Post-filtering only implemented for cepstrum features.")
        if sp_type == "mcep":
            return AudioProcessing.mcep_to_amp_sp(coded_sp, fs, alpha)
        elif sp_type == "mgc":
            return AudioProcessing.mgc_to_amp_sp(coded_sp, fs, alpha, mgc_gamma, n_fft)
        elif sp_type == "amp_sp":
            return coded_sp
        elif sp_type == "mfbanks":
            raise NotImplementedError("sp_type {} is outside the accelerated path.".format(sp_type))
        else:
            raise NotImplementedError("Unknown feature type {}. No decoding method available."
                                      .format(sp_type))

    @staticmethod
    def depreemphasis(raw: np.ndarray, preemphasis: float):
        return scipy.signal.lfilter([1], [1, -preemphasis], raw)
