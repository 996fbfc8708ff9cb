"""WorldFeatLabelGen: the reference's WORLD feature generator / reader surface
(idiaptts/src/data_preparation/world/WorldFeatLabelGen.py) on top of the MI355X-native kernels.

Kept signatures (reference line numbers):
  world_extract_features (:778-807), extract_features (:809-889), trim_to_shortest (:891-907),
  world_features_to_raw (:909-945), convert_to_world_features (:734-762),
  convert_from_world_features (:764-776), _postprocess_world (:357-415), gen_data (:947-1071),
  save_output (:1121-1172), load (:459-573, .npz per stream with the legacy cmp fallback).
What changes underneath: pyworld / pysptk / bandmat calls become batched HIP launches
(idiaptts_amd.world), and gen_data processes `batch_utts` utterances per launch instead of
one utterance at a time (the reference's hot loop #1, :996).
"""
import glob
import logging
import math
import os
from collections import OrderedDict
from typing import List, Union

import numpy as np

from .... import lib as _lib
from .... import parallel as _parallel
from .... import world as _world
from ....misc.mlpg import MLPG
from ....misc.normalisation.MeanCovarianceExtractor import MeanCovarianceExtractor
from ....misc.normalisation.MeanStdDevExtractor import MeanStdDevExtractor
from ....misc.utils import compute_deltas, interpolate_lin  # noqa: F401  (re-exported)
from ..audio.AudioProcessing import AudioProcessing
from ..DataReaderConfig import DataReaderConfig
from ..DataReaders import ReaderBase


_devices_with_tables = set()      # devices on which a gen_data pass of this process has run its first batch


def _dist_device():
    """Device the statistics all-reduce runs on: the current GPU under RCCL ('nccl'), host
    memory under gloo."""
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) \
        if dist.get_backend() == "nccl" else None


def _save_to_npz(file_path, features, feature_name):
    """LabelGen._save_to_npz (reference LabelGen.py:63-101): merge into an existing archive.
    `features` / `feature_name` may be lists: all of them go into the archive in one
    read-modify-write (the reference rewrites the archive once per feature)."""
    os.makedirs(os.path.dirname(file_path), exist_ok=True)
    if not file_path.endswith(".npz"):
        file_path += ".npz"
    if not isinstance(feature_name, (list, tuple)):
        features, feature_name = [features], [feature_name]
    saved = {}
    if os.path.isfile(file_path):
        with np.load(file_path) as archive:
            saved = {k: archive[k] for k in archive.files if k not in feature_name}
    saved.update(zip(feature_name, features))
    tmp = file_path + "_tmp.npz"
    np.savez(tmp, **saved)
    os.replace(tmp, file_path)


def _write_archives_native(cmp_host, f_off, names, dir_out, streams, add_deltas, n_threads):
    """One itts_write_feature_archives call for the utterances of a batch (csrc/hostio.cpp: the
    archives np.savez would write, by a pool of plain threads).  streams: (directory, key,
    (first column, width incl. deltas)).  Returns the (utterance index, stream index) pairs whose
    archive exists with other keys in it: those are merged by the caller (_save_to_npz)."""
    import ctypes
    from .... import lib as _lib
    L = _lib.load()
    n_utts, n_streams = len(names), len(streams)
    paths = (ctypes.c_char_p * (n_utts * n_streams))(*[
        os.fsencode(os.path.join(dir_out, d, os.path.basename(n) + ".npz"))
        for n in names for d, _, _ in streams])
    parts = [3 if (add_deltas and key != WorldFeatLabelGen.ext_vuv) else 1 for _, key, _ in streams]
    col0 = (ctypes.c_int * n_streams)(*[c0 for _, _, (c0, _) in streams])
    width = (ctypes.c_int * n_streams)(*[w // p_ for (_, _, (_, w)), p_ in zip(streams, parts)])
    parts_c = (ctypes.c_int * n_streams)(*parts)
    keys = (ctypes.c_char_p * n_streams)(*[key.encode() for _, key, _ in streams])
    offs = (ctypes.c_int64 * (n_utts + 1))(*[int(o) for o in f_off[:n_utts + 1]])
    flags = (ctypes.c_ubyte * (n_utts * n_streams))()
    _lib.check(L.itts_write_feature_archives(
        cmp_host.ctypes.data, cmp_host.strides[0] // 4, offs, n_utts, paths, n_streams, col0,
        width, parts_c, keys, int(n_threads), ctypes.addressof(flags)),
        "itts_write_feature_archives")
    return [(j // n_streams, j % n_streams) for j, f in enumerate(flags) if f]


class WorldFeatLabelGen(ReaderBase):
    """Create world feat labels for .wav files."""

    class Config(DataReaderConfig):
        """Reader config with the WORLD-specific arguments spelled out (reference :62-135);
        the stream layout is fixed by (sp_type, num_coded_sps, num_bap, add_deltas)."""

        def __init__(self, name, directory=None, features=None, output_names=None,
                     add_deltas=False, preemphasis=0.0, n_fft=None, win_length_ms=None,
                     num_coded_sps=60, num_bap=1, sp_type="mcep", load_sp=True, load_lf0=True,
                     load_vuv=True, load_bap=True, apply_mlpg=True, **kwargs):
            reader_args = {k: kwargs.pop(k) for k in list(kwargs)
                           if k in ("hop_size_ms", "mgc_alpha", "batch_utts")}
            super().__init__(name, WorldFeatLabelGen, directory=directory, features=features,
                             output_names=output_names, **kwargs)
            if type(directory) in (tuple, list):
                directory = directory[0]
            self.directory = directory
            self.kwargs = dict(add_deltas=add_deltas, preemphasis=preemphasis, n_fft=n_fft,
                               win_length_ms=win_length_ms, num_coded_sps=num_coded_sps,
                               num_bap=num_bap, sp_type=sp_type, load_sp=load_sp,
                               load_lf0=load_lf0, load_vuv=load_vuv, load_bap=load_bap,
                               apply_mlpg=apply_mlpg, **reader_args)

    f0_silence_threshold = 30
    lf0_zero = 0
    preemphasis = 0.0
    n_fft = None
    # F0 stage of the extractor: "dio" (DIO + StoneMask = pyworld.wav2world, what the reference
    # runs, :792-793) or "harvest" (pyworld.harvest); an extension, the reference has no switch
    f0_estimator = "dio"
    win_length_ms = None

    dir_lf0 = "lf0"
    dir_vuv = "vuv"
    dir_bap = "bap"
    dir_deltas = "cmp"

    ext_lf0 = "lf0"
    ext_vuv = "vuv"
    ext_bap = "bap"
    ext_deltas = "cmp"

    logger = logging.getLogger(__name__)

    def __init__(self, dir_labels=None, add_deltas=False, preemphasis=0.0, n_fft=None,
                 win_length_ms=None, num_coded_sps=60, num_bap=1, sp_type="mcep", hop_size_ms=5,
                 load_sp=True, load_lf0=True, load_vuv=True, load_bap=True, mgc_alpha=None,
                 batch_utts=32, apply_mlpg=True):
        if type(dir_labels) in (tuple, list):
            dir_labels = dir_labels[0]
        self.dir_labels = dir_labels
        self.apply_mlpg = apply_mlpg
        self._configure("cmp_features", ["acoustic_features"])
        self.add_deltas = add_deltas
        self.preemphasis = preemphasis
        self.n_fft = n_fft
        self.win_length_ms = win_length_ms
        self.num_coded_sps = num_coded_sps
        self.num_bap = num_bap
        self.sp_type = sp_type
        self.hop_size_ms = hop_size_ms
        self.load_sp, self.load_lf0, self.load_vuv, self.load_bap = load_sp, load_lf0, load_vuv, \
            load_bap
        self.mgc_alpha = mgc_alpha  # None: fs_to_mgc_alpha(fs) like the reference
        self.batch_utts = batch_utts
        self.dir_coded_sps = self.sp_type
        if self.num_coded_sps is not None:
            self.dir_coded_sps += str(self.num_coded_sps)
        self.dir_deltas = WorldFeatLabelGen.dir_deltas + "_" + self.dir_coded_sps
        self.covs = [None] * 4
        self.norm_params = None

    # ------------------------------------------------------------------ post-processing / MLPG
    def _postprocess_world(self, sample, norm_params=None, apply_mlpg=True):
        """Turns a de-normalised network output with deltas back into static WORLD features
        (reference :357-415): every continuous stream (coded sp, lf0, bap) goes through MLPG with
        that stream's covariance, V/UV is binarised with `<= 0.5 -> 0` (note:
        convert_to_world_features uses `< 0.5`)."""
        return self._postprocess_world_batch([sample], apply_mlpg=apply_mlpg)[0]

    def _postprocess_world_batch(self, samples, apply_mlpg=True):
        """_postprocess_world for several utterances: one MLPG launch (one host <-> device round
        trip) per stream for all of them instead of one per stream and utterance."""
        if not self.add_deltas:
            return list(samples)
        widths = []  # (stream index into self.covs, static width, is_vuv)
        if self.load_sp:
            widths.append((0, self.num_coded_sps, False))
        if self.load_lf0:
            widths.append((1, 1, False))
        if self.load_vuv:
            widths.append((2, 1, True))
        if self.load_bap:
            widths.append((3, self.num_bap, False))
        pieces = [[] for _ in samples]
        col = 0
        n_cols = samples[0].shape[1]
        # every continuous stream's columns and covariance first: all of them are then ONE upload of the rows,
        # one launch per stream on the device and one download (MLPG.generation_streams)
        jobs, slots = [], []       # (first column, covariance, D); where each piece goes
        for cov_idx, width, is_vuv in widths:
            if is_vuv:
                for u, sample in enumerate(samples):
                    vuv = sample[:, col]      # view: the caller's array is binarised in place,
                    vuv[vuv <= 0.5] = 0.0     # exactly like the reference (:396-399)
                    vuv[vuv > 0.5] = 1.0
                    pieces[u].append(vuv[:, None])
                col += 1
                continue
            c0 = n_cols - width * 3 if cov_idx == 3 else col      # reference slices bap from the END of the row
            col += 3 * width
            if apply_mlpg:
                cov = self.covs[cov_idx]
                jobs.append((c0, cov, cov.shape[0] // 3))
                slots.append(len(pieces[0]))
                for p_ in pieces:
                    p_.append(None)
            else:
                for u, sample in enumerate(samples):
                    pieces[u].append(sample[:, c0:c0 + width])
        if jobs:
            same_width = all(s_.shape[1] == n_cols for s_ in samples)
            if same_width:
                results = MLPG().generation_streams(samples, jobs)
            else:           # (rows of different widths cannot share a matrix: stream by stream)
                results = [MLPG().generation_batch([s_[:, c0:c0 + 3 * d] for s_ in samples], cov, d)
                           for c0, cov, d in jobs]
            for slot, trajs in zip(slots, results):
                for u, traj in enumerate(trajs):
                    pieces[u][slot] = traj
        return [np.concatenate(p_, axis=1) for p_ in pieces]

    # ------------------------------------------------------------------------------ conversions
    @staticmethod
    def convert_to_world_features(sample, contains_deltas=False, num_coded_sps=60, num_bap=1):
        """Slices a [T, (ncs+1+nb)*f + 1] feature matrix into coded_sp, lf0, vuv (`>= 0.5 -> 1`),
        bap; f = 3 with deltas (auto-detected from the width) -- reference :734-762."""
        static = num_coded_sps + 1 + num_bap
        factor = 3 if contains_deltas else 1
        if sample.shape[1] != static * factor + 1:
            if sample.shape[1] != static * 3 + 1:
                raise ValueError("WORLD requires all features to be present.")
            factor = 3
        lf0_col = num_coded_sps * factor
        vuv = (sample[:, lf0_col + factor] >= 0.5).astype(sample.dtype)
        if contains_deltas:
            bap = sample[:, -num_bap * 3:-num_bap * 2]
        else:
            bap = sample[:, -num_bap:]
        return sample[:, :num_coded_sps], sample[:, lf0_col], vuv, bap

    @staticmethod
    def convert_from_world_features(coded_sp, lf0, vuv, bap):
        """reference :764-776"""
        if lf0.ndim < 2:
            lf0 = lf0[:, None]
        if vuv.ndim < 2:
            vuv = vuv[:, None]
        if bap.ndim < 2:
            bap = bap[:, None]
        return np.concatenate((coded_sp, lf0, vuv, bap), axis=1)

    # -------------------------------------------------------------------------------- analysis
    @staticmethod
    def world_extract_features(raw: np.array, fs: int, hop_size_ms: int,
                               f0_silence_threshold: int = None, lf0_zero: float = None,
                               n_fft: int = None):
        """Extract WORLD features (reference :778-807):
        returns amp_sp f64 [T,K], lf0 f32 [T,1], vuv f32 [T,1], bap f32 [T,n_bap]."""
        if f0_silence_threshold is None:
            f0_silence_threshold = WorldFeatLabelGen.f0_silence_threshold
        if lf0_zero is None:
            lf0_zero = WorldFeatLabelGen.lf0_zero
        # one trip to the device: the square root (:795) and lf0 / V-UV (:798-802) are taken there, before the
        # envelope and the contour come back (np.sqrt of 1 321 x 513 values was 0.5 of the call's 2.6 ms)
        res = _world.analyse_batch([np.asarray(raw, dtype=np.float64)], fs, hop_size_ms, n_fft,
                                   want_sp=True, want_bap=True,
                                   f0_method=WorldFeatLabelGen.f0_estimator, amplitude=True,
                                   lf0_params=(f0_silence_threshold, lf0_zero))[0]
        return res["sp"], res["lf0"], res["vuv"], res["bap"]

    @staticmethod
    def extract_features_batch(raws, fs, preemphasis_applied=True, n_fft=None, hop_size_ms=5,
                               sp_type="mcep", num_coded_sps=40, mgc_alpha=None,
                               f0_silence_threshold=None, lf0_zero=None):
        """MI355X-native batched form of extract_features: the spectral envelope never leaves
        the GPU (CheapTrick -> mcep fused). Returns a list of (coded_sp, lf0, vuv, bap)."""
        if sp_type not in ("mcep", "mgc"):
            raise NotImplementedError("Only sp_type 'mcep' and 'mgc' are on the accelerated path.")
        if f0_silence_threshold is None:
            f0_silence_threshold = WorldFeatLabelGen.f0_silence_threshold
        if lf0_zero is None:
            lf0_zero = WorldFeatLabelGen.lf0_zero
        if mgc_alpha is None:
            mgc_alpha = AudioProcessing.fs_to_mgc_alpha(fs)
        cmp_dev, f_off = _world.extract_cmp_batch(
            raws, fs, hop_size_ms, n_fft, num_coded_sps - 1, mgc_alpha, f0_silence_threshold,
            lf0_zero, add_deltas=False,
            mgc_gamma=AudioProcessing.mgc_gamma if sp_type == "mgc" else None,
            f0_method=WorldFeatLabelGen.f0_estimator)
        cmp_host = cmp_dev.cpu().numpy()
        out = []
        for u in range(len(raws)):
            c = cmp_host[f_off[u]:f_off[u + 1]]
            out.append((np.ascontiguousarray(c[:, :num_coded_sps]),
                        np.ascontiguousarray(c[:, num_coded_sps:num_coded_sps + 1]),
                        np.ascontiguousarray(c[:, num_coded_sps + 1:num_coded_sps + 2]),
                        np.ascontiguousarray(c[:, num_coded_sps + 2:])))
        return out

    @staticmethod
    def extract_features(dir_in, file_name: str, file_ext: str = "wav", preemphasis: float = 0.0,
                         n_fft: int = None, win_length_ms: int = None, hop_size_ms: int = 5,
                         sp_type: str = "mcep", num_coded_sps: int = 40, load_sp: bool = True,
                         load_lf0: bool = True, load_vuv: bool = True, load_bap: bool = True,
                         f0_silence_threshold: int = None, lf0_zero: float = None,
                         mgc_alpha: float = None):
        """Extract acoustic features from a single audio file (reference :809-889)."""
        audio_name = os.path.join(dir_in, file_name + "." + file_ext)
        raw, fs = AudioProcessing.get_raw(audio_name, preemphasis)
        coded_sp, lf0, vuv, bap = WorldFeatLabelGen.extract_features_batch(
            [raw], fs, n_fft=n_fft, hop_size_ms=hop_size_ms, sp_type=sp_type,
            num_coded_sps=num_coded_sps, mgc_alpha=mgc_alpha,
            f0_silence_threshold=f0_silence_threshold, lf0_zero=lf0_zero)[0]
        if load_vuv:
            unvoiced_frames_percentage = vuv.sum() / len(vuv) * 100.0
            if unvoiced_frames_percentage < 5.0:
                logging.warning("Detected only {:.0f}% [{}/{}] unvoiced frames in {}.".format(
                    unvoiced_frames_percentage, int(vuv.sum()), len(vuv), file_name))
        logging.info("Extracted (WORLD {}{}, lf0, vuv, {}bap) features from {} at {} Hz with {} ms"
                     " frame hop.".format(coded_sp.shape[1], sp_type, bap.shape[1],
                                          os.path.basename(file_name), fs, hop_size_ms))
        return coded_sp if load_sp else None, lf0, vuv, bap

    @staticmethod
    def trim(sample, trim_width):
        idx = [slice(v[0], sample.shape[dim] - v[1]) if isinstance(v, tuple) else v
               for dim, v in enumerate(trim_width)]
        return sample[tuple(idx)]

    @staticmethod
    def trim_to_shortest(features):
        """reference :891-907 (front = diff // 2, end = diff - front)"""
        len_shortest = min(map(len, [f for f in features if f is not None]))
        for idx, feature in enumerate(features):
            if feature is None:
                continue
            len_diff = len(feature) - len_shortest
            if len_diff > 0:
                trim_front = len_diff // 2
                trim_end = len_diff - trim_front
                features[idx] = WorldFeatLabelGen.trim(feature, ((trim_front, trim_end),))
        return features

    # ------------------------------------------------------------------------------- synthesis
    @staticmethod
    def world_features_to_raw(amp_sp: np.array, lf0: np.array, vuv: np.array, bap: np.array,
                              fs: int, n_fft: int = None, f0_silence_threshold: int = None,
                              lf0_zero: float = None, preemphasis: float = 0.0):
        """WORLD vocoder waveform generation (reference :909-945). Mutates `vuv` in place like
        the reference (:930)."""
        return WorldFeatLabelGen.world_features_to_raw_batch(
            [amp_sp], [lf0], [vuv], [bap], fs, n_fft, f0_silence_threshold, lf0_zero,
            preemphasis)[0]

    @staticmethod
    def world_features_to_raw_batch(amp_sps, lf0s, vuvs, baps, fs, n_fft=None,
                                    f0_silence_threshold=None, lf0_zero=None, preemphasis=0.0):
        if f0_silence_threshold is None:
            f0_silence_threshold = WorldFeatLabelGen.f0_silence_threshold
        if lf0_zero is None:
            lf0_zero = WorldFeatLabelGen.lf0_zero
        if n_fft is None:
            n_fft = AudioProcessing.fs_to_frame_length(fs)
        f0s, sps, bps = [], [], []
        for amp_sp, lf0, vuv, bap in zip(amp_sps, lf0s, vuvs, baps):
            # (pow_sp = np.square(amp_sp, dtype=np.float64) of the reference, :925, is taken on the device after the
            # upload: the same IEEE products, 0.3 ms of host time less per utterance)
            pow_sp = np.asarray(amp_sp)
            if pow_sp.dtype not in (np.float32, np.float64):
                pow_sp = pow_sp.astype(np.float64)
            f0 = np.exp(lf0, dtype=np.float64)
            vuv[f0 < f0_silence_threshold] = 0  # WORLD throws an error for too small f0 values.
            f0[vuv == 0] = lf0_zero
            if f0.ndim > 1:
                assert f0.shape[1:] == (1,) * (f0.ndim - 1), \
                    "F0 should have only one dimension at this stage."
                f0 = f0.squeeze()
            if bap.ndim < 2:
                bap = bap.reshape(-1, 1)
            f0s.append(np.atleast_1d(f0))
            sps.append(pow_sp)
            bps.append(np.ascontiguousarray(bap, np.float64))
        return _world.synthesise_batch(f0s, sps, bps, fs, n_fft, 5.0, preemphasis, sp_is_amplitude=True)

    # ---------------------------------------------------------------------------------- gen_data
    def _create_norm_params_extractors(self):
        cls = MeanCovarianceExtractor if self.add_deltas else MeanStdDevExtractor
        self.norm_params_ext_coded_sp = cls()
        self.norm_params_ext_lf0 = cls()
        self.norm_params_ext_bap = cls()

        class NormaliserVUVDummy(object):
            def add_sample(self, *args):
                pass

            def save(self, *args):
                pass

            def get_params(self):
                return (0.0,), (1.0,)
        self.norm_params_ext_vuv = NormaliserVUVDummy()

    def _streams(self):
        return list(zip((self.load_sp, self.load_lf0, self.load_vuv, self.load_bap),
                        (self.dir_coded_sps, self.dir_lf0, self.dir_vuv, self.dir_bap),
                        (self.sp_type, self.ext_lf0, self.ext_vuv, self.ext_bap),
                        (self.norm_params_ext_coded_sp, self.norm_params_ext_lf0,
                         self.norm_params_ext_vuv, self.norm_params_ext_bap)))

    def save_output(self, features, dir_out, file_name):
        """reference :1121-1172"""
        output = list()
        for (load, feature_dir, feature_ext, normaliser), feature in zip(self._streams(),
                                                                         features):
            if not load:
                continue
            file_name = os.path.basename(file_name)
            if self.add_deltas:
                if feature_dir != self.dir_vuv:
                    deltas = compute_deltas(feature)
                    double_deltas = compute_deltas(deltas)
                if dir_out is not None:
                    file_path = os.path.join(dir_out, feature_dir, file_name)
                    if feature_dir != self.dir_vuv:
                        _save_to_npz(file_path, [feature, deltas, double_deltas],
                                     [feature_ext, feature_ext + "_deltas",
                                      feature_ext + "_double_deltas"])
                    else:
                        _save_to_npz(file_path, feature, feature_ext)
                if feature_dir != self.dir_vuv:
                    feature = np.concatenate((feature, deltas, double_deltas), axis=1)
            elif dir_out is not None:
                _save_to_npz(os.path.join(dir_out, feature_dir, file_name), feature, feature_ext)
            normaliser.add_sample(feature)
            output.append(feature)
        return output

    # stream layout of the feature matrix the device assembles (itts_assemble_cmp_f32)
    def _cmp_columns(self, num_bap=None):
        f = 3 if self.add_deltas else 1
        ncs, nb = self.num_coded_sps, self.num_bap if num_bap is None else num_bap
        return {"sp": (0, f * ncs), "lf0": (f * ncs, f), "vuv": (f * (ncs + 1), 1),
                "bap": (f * (ncs + 1) + 1, f * nb)}

    def _write_utterance(self, dir_out, name, cmp_u, cols):
        """The files save_output writes for one utterance (:1121-1172), from its feature matrix."""
        for (load, feature_dir, ext, _), key in zip(self._streams(), ("sp", "lf0", "vuv", "bap")):
            if not load:
                continue
            c0, w = cols[key]
            path = os.path.join(dir_out, feature_dir, name)
            if self.add_deltas and key != "vuv":
                k = w // 3
                _save_to_npz(path, [np.ascontiguousarray(cmp_u[:, c0 + i * k:c0 + (i + 1) * k])
                                    for i in range(3)],
                             [ext, ext + "_deltas", ext + "_double_deltas"])
            else:
                _save_to_npz(path, np.ascontiguousarray(cmp_u[:, c0:c0 + w]), ext)

    def _batch_schedule(self, id_list):
        """The id list cut into the batches the device analyses.  `batch_utts` is the size of a batch in the middle of
        the list; ITTS_GEN_DATA_SCHEDULE="32,96,128" (lab) gives explicit sizes for the first batches, mirrored at the
        end, with `batch_utts` in between."""
        n = len(id_list)
        sizes = []
        lab = os.environ.get("ITTS_GEN_DATA_SCHEDULE")
        if lab:
            head = [int(v) for v in lab.split(",") if v.strip()]
            tail = head[::-1]
            left = n
            front = []
            for v in head:
                if left - sum(tail) <= 0 or v >= left:
                    break
                front.append(v)
                left -= v
            back = []
            for v in tail[::-1][:len(front)][::-1]:
                if v < left:
                    back.append(v)
                    left -= v
            while left > 0:
                v = min(self.batch_utts, left)
                front.append(v)
                left -= v
            sizes = front + back
        else:
            sizes = [min(self.batch_utts, n - b0) for b0 in range(0, n, self.batch_utts)]
        out, b0 = [], 0
        for v in sizes:
            out.append(id_list[b0:b0 + v])
            b0 += v
        assert b0 == n
        return out

    def _gen_data_pipeline(self, dir_in, dir_out, file_ext, id_list, label_dict):
        """The hot loop of gen_data (reference :996-1013, one utterance at a time on one core):
        reader threads decode and pre-emphasise the next batches, the device turns a batch of
        waveforms into the finished feature matrix (analysis, lf0 interpolation, deltas, stream
        layout) and adds it to the normalisation sums, ONE device -> host copy per batch, writer
        threads store the per-utterance archives while the next batch is analysed."""
        import concurrent.futures as cf
        import torch
        loaded = [k for k, load in zip(("sp", "lf0", "vuv", "bap"), self.load_flags) if load]
        cols = stats = None
        batches = self._batch_schedule(id_list)

        n_io = max(2, min(16, (os.cpu_count() or 2) // 2))

        upload_stream = torch.cuda.Stream()

        def read(names):
            # decoded into page-locked memory and sent to the device from the reader thread, on a stream of its own:
            # the 51 MB of a 64-utterance batch arrive while the batch before is analysed
            samples, x_off, fss = AudioProcessing.get_raw_batch(
                [os.path.join(dir_in, n + "." + file_ext) for n in names], self.preemphasis, n_io, pinned=True)
            arrived = None
            if isinstance(samples, torch.Tensor):
                with torch.cuda.stream(upload_stream):
                    on_device = samples.to("cuda", non_blocking=True)
                    arrived = upload_stream.record_event()
                samples = on_device
            return samples, x_off, fss, arrived

        def write(names, cmp_host, f_off, cols):
            streams = [(d, ext, cols[k]) for (load, d, ext, _), k
                       in zip(self._streams(), ("sp", "lf0", "vuv", "bap")) if load]
            merge = _write_archives_native(cmp_host, f_off, names, dir_out, streams,
                                           self.add_deltas, max(2, n_io // 2))
            mark("written: " + names[0])
            for u, si in merge:      # archive with foreign keys: _save_to_npz keeps them
                d, ext, (c0, w) = streams[si]
                cmp_u = cmp_host[f_off[u]:f_off[u + 1]]
                path = os.path.join(dir_out, d, os.path.basename(names[u]))
                if self.add_deltas and ext != self.ext_vuv:
                    k = w // 3
                    _save_to_npz(path, [np.ascontiguousarray(cmp_u[:, c0 + i * k:c0 + (i + 1) * k])
                                        for i in range(3)],
                                 [ext, ext + "_deltas", ext + "_double_deltas"])
                else:
                    _save_to_npz(path, np.ascontiguousarray(cmp_u[:, c0:c0 + w]), ext)

        # Three stages in flight: batch b + 1 .. b + 3 are being read, batch b is analysed (the host drives its Newton
        # rounds), batch b - 1 goes device -> host on a stream of its own and is handed to the writers when the NEXT
        # batch's analysis has been queued -- so neither the 57-MB copy of a 64-utterance batch nor a wait for it sits
        # between two analyses (round 4: copy + synchronize per batch, 2.3 ms of every 17).
        depth = 3
        copy_stream = torch.cuda.Stream()
        # writer jobs are chunks of at most `chunk` utterances, four at a time: a batch's archives leave in parallel
        # pieces, and the last batch -- whose write nothing overlaps -- is not one 13-ms job behind two others
        chunk = max(8, int(os.environ.get("ITTS_GEN_DATA_WRITE_CHUNK", "32")))
        n_writers = max(1, int(os.environ.get("ITTS_GEN_DATA_WRITERS", "4")))
        with cf.ThreadPoolExecutor(depth) as readers, cf.ThreadPoolExecutor(n_writers) as writers:
            pending_reads = [readers.submit(read, names) for names in batches[:depth]]
            writes = []
            trace = os.environ.get("ITTS_GEN_DATA_TRACE") == "1"
            import time as _time
            t_pass = _time.perf_counter()
            events = []

            def mark(what, bi=-1):
                if trace:
                    events.append(((_time.perf_counter() - t_pass) * 1e3, what, bi))
            in_copy = None          # (names, host tensor, f_off, event) of the batch whose copy is in flight

            def hand_over(item):
                names_, host_, f_off_, done_ = item
                done_.synchronize()
                mark("copy to host done: " + names_[0])
                cmp_host = host_.numpy()
                if dir_out is not None:
                    for a in range(0, len(names_), chunk):
                        writes.append(writers.submit(write, names_[a:a + chunk], cmp_host,
                                                     f_off_[a:a + chunk + 1], cols))
                if label_dict is not None:
                    for u, n in enumerate(names_):
                        cmp_u = cmp_host[f_off_[u]:f_off_[u + 1]]
                        label_dict[n] = np.concatenate(
                            [cmp_u[:, cols[k][0]:cols[k][0] + cols[k][1]] for k in loaded], axis=1) \
                            if loaded else None

            # The analysis of a batch is driven by the host (the Newton rounds of the mel-cepstrum read a count back
            # between launches) and a 64-utterance batch leaves the chip partly idle at the tails of its kernels: two
            # batches are analysed at a time, by two threads on two streams (the library's calls release the
            # interpreter lock; its scratch blocks and tables are stream-ordered and behind mutexes).  Results are
            # taken in batch order: the statistics add up in the order they always did.
            n_flight = max(1, int(os.environ.get("ITTS_GEN_DATA_INFLIGHT", "2")))
            analysis_streams = [torch.cuda.Stream() for _ in range(n_flight)]

            def analyse(bi):
                mark("analysis thread starts", bi)
                samples, x_off, fss, arrived = pending_reads[bi].result()
                mark("read there", bi)
                pending_reads[bi] = None
                assert len(set(fss)) == 1, "All files of a batch need the same sampling rate."
                fs = fss[0]
                alpha = self.mgc_alpha if self.mgc_alpha is not None \
                    else AudioProcessing.fs_to_mgc_alpha(fs)
                st = analysis_streams[bi % n_flight]
                with torch.cuda.stream(st):
                    if arrived is not None:
                        st.wait_event(arrived)
                        samples.record_stream(st)
                    cmp_dev, f_off = _world.extract_cmp_batch(
                        (samples, x_off), fs, self.hop_size_ms, self.n_fft, self.num_coded_sps - 1,
                        alpha, WorldFeatLabelGen.f0_silence_threshold, WorldFeatLabelGen.lf0_zero,
                        self.add_deltas,
                        mgc_gamma=AudioProcessing.mgc_gamma if self.sp_type == "mgc" else None,
                        f0_method=self.f0_estimator)
                    ready = st.record_event()
                mark("analysis queued (host side done)", bi)
                return cmp_dev, f_off, ready, fs

            with cf.ThreadPoolExecutor(n_flight) as analysers:
                analyses = {}

                def start(bi):
                    analyses[bi] = analysers.submit(analyse, bi)
                    if bi + depth < len(batches):
                        pending_reads.append(readers.submit(read, batches[bi + depth]))

                for bi, names in enumerate(batches):
                    t_a = _time.perf_counter()
                    if bi not in analyses:
                        start(bi)
                    if bi == 0 and torch.cuda.current_device() not in _devices_with_tables:
                        # the first batch of the process alone: it builds the device's tables (on its stream); later
                        # calls find them there and start two batches at once like every other pair (a lone first
                        # batch was 15 of a 100-ms pass over 512 utterances)
                        analyses[0].result()
                        torch.cuda.synchronize()
                        _devices_with_tables.add(torch.cuda.current_device())
                    for ahead in range(1, n_flight):
                        if bi + ahead < len(batches) and bi + ahead not in analyses:
                            start(bi + ahead)
                    cmp_dev, f_off, ready, fs = analyses.pop(bi).result()
                    mark("main has the result", bi)
                    t_b = _time.perf_counter()
                    main = torch.cuda.current_stream()
                    main.wait_event(ready)
                    cmp_dev.record_stream(main)
                    if cols is None:      # WORLD fixes the number of bap bands by the sampling rate
                        cols = self._cmp_columns(AudioProcessing.fs_to_num_bap(fs))
                        stats = _world.StreamStats({k: cols[k] for k in loaded if k != "vuv"},
                                                   self.add_deltas)
                    stats.add(cmp_dev)
                    ready = main.record_event()
                    host = torch.empty(cmp_dev.shape, dtype=torch.float32, pin_memory=True)
                    with torch.cuda.stream(copy_stream):
                        copy_stream.wait_event(ready)
                        host.copy_(cmp_dev, non_blocking=True)
                        done = copy_stream.record_event()
                    cmp_dev.record_stream(copy_stream)
                    if in_copy is not None:
                        hand_over(in_copy)      # the batch before: its copy ran beside this batch's analysis
                    in_copy = (names, host, f_off, done)
                    if trace:
                        print("gen_data batch {}: waited {:.1f} ms for its analysis".format(bi, (t_b - t_a) * 1e3))
            if in_copy is not None:
                hand_over(in_copy)
            t_a = _time.perf_counter()
            for w in writes:
                w.result()          # re-raises a writer's exception
            if trace:
                print("gen_data: waited {:.1f} ms for the writers".format(
                    (_time.perf_counter() - t_a) * 1e3))
                mark("pass over")
                for t, what, bi in sorted(events):
                    print("  {:8.1f} ms  {}{}".format(t, what, "" if bi < 0 else " [batch {}]".format(bi)))
        for (load, _, _, normaliser), key in zip(self._streams(), ("sp", "lf0", "vuv", "bap")):
            if load and key != "vuv" and stats is not None:
                stats.store(key, normaliser)

    def gen_data(self, dir_in, dir_out=None, file_id_list="", file_ext="wav", id_list=None,
                 return_dict=False):
        """Prepare acoustic features from audio files (reference :947-1071); utterances are
        analysed `batch_utts` at a time on the GPU.

        Under torch.distributed (one process per GPU) the utterance list is partitioned over the
        ranks, length balanced by file size, every rank writes the features of its own utterances
        and the additive normalisation statistics are merged with one sum all-reduce per stream
        (SURVEY.md section 8e); all ranks return the same parameters, rank 0 writes them."""
        if id_list is None:
            id_list = [os.path.splitext(os.path.basename(f))[0]
                       for f in glob.glob(os.path.join(dir_in, "*" + file_ext))]
            file_id_list_name = "all"
        else:
            file_id_list_name = os.path.splitext(os.path.basename(file_id_list))[0]
        if dir_out is not None:
            for load, d, _, _ in [(self.load_sp, self.dir_coded_sps, 0, 0),
                                  (self.load_lf0, self.dir_lf0, 0, 0),
                                  (self.load_vuv, self.dir_vuv, 0, 0),
                                  (self.load_bap, self.dir_bap, 0, 0)]:
                if load:
                    os.makedirs(os.path.join(dir_out, d), exist_ok=True)
        label_dict = OrderedDict()
        self._create_norm_params_extractors()
        rank, world = _parallel.dp_rank_world()
        all_ids = list(id_list)
        if world > 1:
            sizes = [os.path.getsize(os.path.join(dir_in, n + "." + file_ext)) for n in all_ids]
            id_list = [all_ids[i] for i in _parallel.shard_by_length(sizes, world)[rank]]
        if self.sp_type not in ("mcep", "mgc"):
            raise NotImplementedError("Only sp_type 'mcep' and 'mgc' are on the accelerated path.")
        self._gen_data_pipeline(dir_in, dir_out, file_ext, id_list, label_dict if return_dict
                                else None)
        if world > 1:
            import torch.distributed as dist
            dev = _dist_device()
            for load, _, ext, normaliser in self._streams():
                if load and ext != self.ext_vuv:
                    _parallel.allreduce_stats_(normaliser, device=dev)
            if return_dict:
                gathered = [None] * world
                dist.all_gather_object(gathered, label_dict)
                merged = {}
                for d in gathered:
                    merged.update(d)
                label_dict = OrderedDict((n, merged[n]) for n in all_ids)
        output_means, output_std_dev = list(), list()
        for load, feature_dir, ext, normaliser in self._streams():
            if not load:
                continue
            norm = normaliser.get_params()
            output_means.append(norm[0])
            output_std_dev.append(norm[1])
            if dir_out and rank == 0:
                norm_file_path = os.path.join(dir_out, feature_dir, file_id_list_name)
                if self.add_deltas and ext != self.ext_vuv:
                    if file_id_list_name is not None and os.path.basename(file_id_list_name) != "":
                        norm_file_path += "-"
                    norm_file_path += "deltas"
                normaliser.save(norm_file_path)
        if not self.add_deltas:
            if len(output_means) > 0:
                output_means = np.concatenate(output_means, axis=0)
                output_std_dev = np.concatenate(output_std_dev, axis=0)
            else:
                output_means = output_std_dev = None
        if world > 1:
            dist.barrier()          # the parameter files exist when any rank returns
        if return_dict:
            return label_dict, output_means, output_std_dev
        return output_means, output_std_dev

    # ----------------------------------------------------------------------------- reader surface
    @property
    def load_flags(self):
        return (self.load_sp, self.load_lf0, self.load_vuv, self.load_bap)

    def get_normalisation_params(self, dir_out=None, file_name=None):
        """(mean, std_dev) over the loaded streams, V/UV as (0, 1); with deltas the per-stream
        covariances land in self.covs for MLPG (reference :575-731).  Looks for the per-stream
        `.npz` files gen_data writes (`<dir>/<stream>/[<name>-]deltas-mean-covariance.npz` or
        `[<name>-]mean-std_dev.npz`), then for the legacy `<dir>/cmp_<sp><n>/[<name>-]<stream>-
        mean-covariance.bin` and its older `mean-covariance_<stream>.bin` spelling."""
        if dir_out is None:
            dir_out = self.dir_labels
        sub_dirs = (self.dir_coded_sps, self.dir_lf0, self.dir_vuv, self.dir_bap)
        prefix = "" if file_name is None or os.path.basename(file_name) == "" \
            else file_name + "-"
        means, std_devs = [], []
        try:
            for idx, (load, sub) in enumerate(zip(self.load_flags, sub_dirs)):
                if not load:
                    continue
                if sub == self.dir_vuv:
                    means.append(np.atleast_2d(0.0))
                    std_devs.append(np.atleast_2d(1.0))
                    continue
                base = os.path.join(dir_out, sub, prefix)
                if self.add_deltas:
                    mean, cov, std_dev = MeanCovarianceExtractor.load(
                        base + "deltas-" + MeanCovarianceExtractor.file_name_appendix + ".npz")
                    self.covs[idx] = cov
                else:
                    mean, std_dev = MeanStdDevExtractor.load(
                        base + MeanStdDevExtractor.file_name_appendix + ".npz")
                means.append(np.atleast_2d(mean))
                std_devs.append(np.atleast_2d(std_dev))
            self.norm_params = (np.concatenate(means, axis=1), np.concatenate(std_devs, axis=1))
            return self.norm_params
        except FileNotFoundError as e0:
            means, std_devs = [], []
            for idx, (load, sub) in enumerate(zip(self.load_flags, sub_dirs)):
                if not load:
                    continue
                if sub == self.dir_vuv:
                    means.append(np.atleast_2d(0.0))
                    std_devs.append(np.atleast_2d(1.0))
                    continue
                appendix = MeanCovarianceExtractor.file_name_appendix
                candidates = [os.path.join(dir_out, self.dir_deltas,
                                           "{}{}-{}.bin".format(prefix, sub, appendix)),
                              os.path.join(dir_out, self.dir_deltas,
                                           "{}{}_{}.bin".format(prefix, appendix, sub))]
                for path in candidates:
                    if os.path.isfile(path):
                        mean, cov, std_dev = MeanCovarianceExtractor.load(path)
                        break
                else:
                    raise FileNotFoundError([e0] + candidates)
                if not self.add_deltas:
                    n = len(cov) // 3
                    assert len(cov) == 3 * n, "Legacy statistics are expected to hold deltas."
                    cov, mean, std_dev = cov[:n, :n], mean[:n], std_dev[:n]
                self.covs[idx] = cov
                means.append(np.atleast_2d(mean))
                std_devs.append(np.atleast_2d(std_dev))
            self.norm_params = (np.concatenate(means, axis=1), np.concatenate(std_devs, axis=1))
            if self.add_deltas:
                self.norm_params = (self.norm_params[0][0], self.norm_params[1][0])
            return self.norm_params

    def preprocess_sample(self, sample, norm_params=None):
        """(x - mean) / std_dev, float32 (NpzDataReader.preprocess_sample :347-371)."""
        mean, std_dev = self.norm_params if norm_params is None else norm_params
        return _lib.normalise_rows(sample, mean, std_dev)

    def postprocess_sample(self, sample, norm_params=None, apply_mlpg=None):
        """De-normalise, then MLPG per stream (reference :338-355 / NpzDataReader :399-420)."""
        mean, std_dev = self.norm_params if norm_params is None else norm_params
        sample = sample * std_dev + mean
        return self._postprocess_world(
            sample, apply_mlpg=self.apply_mlpg if apply_mlpg is None else apply_mlpg)

    def postprocess_sample_batch(self, samples, norm_params=None, apply_mlpg=None):
        """[postprocess_sample(s) for s in samples] with the MLPG solves of all utterances batched
        per stream (what the trainers' forward / synth use for a mini-batch of outputs)."""
        mean, std_dev = self.norm_params if norm_params is None else norm_params
        return self._postprocess_world_batch(
            [sample * std_dev + mean for sample in samples],
            apply_mlpg=self.apply_mlpg if apply_mlpg is None else apply_mlpg)

    @staticmethod
    def load_sample(id_name, dir_out, add_deltas=False, num_coded_sps=60, num_bap=1,
                    sp_type="mcep", load_sp=True, load_lf0=True, load_vuv=True, load_bap=True):
        """Un-normalised features of one id (reference :417-457)."""
        assert dir_out is not None, "dir_out cannot be None"
        id_name = os.path.splitext(os.path.basename(id_name))[0]
        return WorldFeatLabelGen(dir_labels=dir_out, add_deltas=add_deltas,
                                 num_coded_sps=num_coded_sps, num_bap=num_bap, sp_type=sp_type,
                                 load_sp=load_sp, load_lf0=load_lf0, load_vuv=load_vuv,
                                 load_bap=load_bap).load(id_name)

    # -------------------------------------------------------------------------------------- load
    def load(self, id_name: str):
        """Per-stream .npz (keys '<ext>[_deltas|_double_deltas]') with the legacy cmp fallback
        (reference :459-573)."""
        deltas_factor = 3 if self.add_deltas else 1
        dim_coded_sp = self.num_coded_sps * deltas_factor
        dim_lf0 = 1 * deltas_factor
        dim_vuv = 1
        dim_bap = self.num_bap * deltas_factor
        id_name = os.path.basename(id_name)
        try:
            output_list = list()
            for load, feature_dir, ext in zip(
                    (self.load_sp, self.load_lf0, self.load_vuv, self.load_bap),
                    (self.dir_coded_sps, self.dir_lf0, self.dir_vuv, self.dir_bap),
                    (self.sp_type, self.ext_lf0, self.ext_vuv, self.ext_bap)):
                if not load:
                    continue
                archive = np.load(os.path.join(self.dir_labels, feature_dir, id_name + ".npz"))
                labels = archive[ext]
                if self.add_deltas and ext != self.ext_vuv:
                    labels = np.concatenate((labels, archive[ext + "_deltas"],
                                             archive[ext + "_double_deltas"]), axis=1)
                output_list.append(labels)
        except FileNotFoundError:
            output_list = list()
            path = os.path.join(self.dir_labels,
                                "{}_{}{}".format(WorldFeatLabelGen.dir_deltas, self.sp_type,
                                                 self.num_coded_sps),
                                "{}.{}".format(id_name, WorldFeatLabelGen.ext_deltas))
            cmp_ = np.fromfile(path, dtype=np.float32)
            total_dims = 3 * (self.num_coded_sps + 1 + self.num_bap) + dim_vuv
            labels = np.reshape(cmp_, [-1, total_dims])
            if self.load_sp:
                output_list.append(labels[:, :dim_coded_sp])
            if self.load_lf0:
                s = 3 * self.num_coded_sps
                output_list.append(labels[:, s:s + dim_lf0])
            if self.load_vuv:
                s = -3 * self.num_bap - dim_vuv
                output_list.append(labels[:, s:-3 * self.num_bap])
            if self.load_bap:
                if dim_bap == 3 * self.num_bap:
                    output_list.append(labels[:, -3 * self.num_bap:])
                else:
                    s = -3 * self.num_bap
                    output_list.append(labels[:, s:s + dim_bap])
        assert len(output_list) > 0, "At least one type of acoustic feature has to be loaded."
        return np.concatenate(output_list, axis=1)
