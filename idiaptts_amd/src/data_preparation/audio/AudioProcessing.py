"""AudioProcessing with the reference's interface for the WORLD hot path
(idiaptts/src/data_preparation/audio/AudioProcessing.py); pysptk / pyworld calls are replaced
by the HIP kernels behind libidiaptts_amd.so. librosa-based helpers (mel filter banks,
Griffin-Lim) are out of scope."""
import logging

import numpy as np
import scipy.io.wavfile
import scipy.signal
import torch

from .... import lib as _lib
from .... import ops


def _dev():
    _lib.require_gpu()
    return torch.device("cuda")


class AudioProcessing:
    mgc_gamma = -1. / 3.

    @staticmethod
    def fs_to_mgc_alpha(fs):
        """pysptk.util.mcepalpha(fs) (reference :32-40)."""
        return _lib.load().itts_mcep_alpha(int(fs))

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
    def extract_mcep(amp_sp: np.array, num_coded_sps: int, mgc_alpha: float) -> np.array:
        """pysptk.mcep(amp_sp, order, alpha, eps=1e-8, min_det=0, etype=1, itype=3) as float32
        (reference :142-153)."""
        a = torch.from_numpy(np.ascontiguousarray(amp_sp, dtype=np.float64)).to(_dev())
        return ops.mcep(a, num_coded_sps - 1, mgc_alpha, eps=1.0e-8).cpu().numpy()

    @staticmethod
    def extract_mgc(amp_sp, fs=None, num_coded_sps=60, mgc_alpha=None):
        raise NotImplementedError("mgcep (gamma=-1/3) is a 'next' row (SURVEY.md section 8f); "
                                  "use sp_type='mcep'.")

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
            logging.warning("merlin_post_filter is not part of the hot path; ignoring.")
        if sp_type == "mcep":
            return AudioProcessing.mcep_to_amp_sp(coded_sp, fs, alpha)
        elif sp_type == "amp_sp":
            return coded_sp
        elif sp_type in ("mgc", "mfbanks"):
            raise NotImplementedError("sp_type {} is outside the accelerated path.".format(sp_type))
        else:
            raise NotImplementedError("Unknown feature type {}. No decoding method available."
                                      .format(sp_type))

    @staticmethod
    def depreemphasis(raw: np.ndarray, preemphasis: float):
        return scipy.signal.lfilter([1], [1, -preemphasis], raw)
