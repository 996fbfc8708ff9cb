"""Batched, GPU-resident WORLD feature path (the MI355X-native fast path behind the drop-in shims).

analysis : wav(s) -> DIO -> StoneMask -> CheapTrick (+ fused SPTK mcep) / D4C (+ coded bap)
synthesis: (f0, sp, ap) -> WORLD synthesis -> float32 (+ de-pre-emphasis)
Utterances are concatenated and processed by single launches; only the small per-frame features
(f0, mcep, bap) travel back to the host unless the spectral envelope is asked for.
"""
import numpy as np
import torch

from . import lib as _lib
from . import ops


def _device(device):
    _lib.require_gpu()
    return torch.device(device if device is not None else "cuda")


_side_streams = {}


def _side_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _side_streams:
        # (stream priorities: this stack has two levels, default and high -- measured, the side stream high or not
        # makes no difference to where the two streams' kernels end: scripts/r5_job36.sh)
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


def num_frames(n, fs, hop_ms=5.0):
    return int(_lib.load().itts_world_num_frames(int(n), int(fs), float(hop_ms)))


def offsets(lengths):
    off = [0]
    for n in lengths:
        off.append(off[-1] + int(n))
    return off


def lf0_vuv_from_f0(f0, f0_silence_threshold=30, lf0_zero=0, device=None):
    """WorldFeatLabelGen.world_extract_features :798-802 (float32 log, threshold,
    interpolate_lin) for one contour: (lf0 [T, 1] f32, vuv [T, 1] f32)."""
    dev = _device(device)
    f0 = np.ascontiguousarray(f0, dtype=np.float64).reshape(-1)
    lf0, vuv = ops.lf0_vuv(torch.from_numpy(f0).to(dev), [0, len(f0)], f0_silence_threshold,
                           lf0_zero)
    return lf0.cpu().numpy()[:, None], vuv.cpu().numpy()[:, None]


def estimate_f0(x, x_off, f_off, fs, hop_ms=5.0, f0_method="dio"):
    """The F0 stage: "dio" = pyworld.wav2world's DIO + StoneMask (what the reference extracts with,
    WorldFeatLabelGen.py:792-793); "harvest" = pyworld.harvest (no StoneMask pass: Harvest refines
    its own candidates).  Both share the frame grid int(1000 n / fs / hop) + 1."""
    if f0_method == "dio":
        return ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs, hop_ms), f_off, fs, hop_ms)
    if f0_method == "harvest":
        return ops.harvest(x, x_off, f_off, fs, hop_ms)
    raise NotImplementedError("Unknown F0 estimator {} (dio, harvest).".format(f0_method))


def analyse_batch(raws, fs, hop_ms=5.0, n_fft=None, want_sp=True, want_ap=False,
                  mcep_order=None, mcep_alpha=None, want_bap=True, device=None, f0_method="dio",
                  amplitude=False, lf0_params=None):
    """raws: list of float64 waveforms (already pre-emphasised). Returns a list of dicts with
    f0 [T] f64, and optionally sp [T,K] f64 (power; with `amplitude` its square root, taken on the device),
    ap [T,K] f64, mcep [T,order+1] f32, bap [T,nap] f32; with lf0_params = (f0_silence_threshold, lf0_zero)
    also lf0 / vuv [T, 1] f32 (WorldFeatLabelGen.py:798-802) from the contour while it is still on the device."""
    dev = _device(device)
    L = _lib.load()
    n_fft = n_fft or L.itts_cheaptrick_fft_size(int(fs), 71.0)
    x_off = offsets([len(r) for r in raws])
    f_off = offsets([num_frames(len(r), fs, hop_ms) for r in raws])
    x = torch.from_numpy(np.ascontiguousarray(np.concatenate(raws), dtype=np.float64)).to(dev)
    f0 = estimate_f0(x, x_off, f_off, fs, hop_ms, f0_method)
    sp = mc = ap = bap = None
    # CheapTrick/mcep and D4C only share their inputs: run D4C on a side stream so the two
    # occupancy-bound kernels overlap
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    if want_ap or want_bap:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ap, bap = ops.d4c(x, x_off, f0, f_off, fs, hop_ms, n_fft, want_ap=want_ap,
                              want_bap=torch.float32 if want_bap else None)
    if want_sp or mcep_order is not None:
        sp, mc, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, hop_ms, n_fft, want_sp=want_sp,
                                        order=mcep_order, alpha=mcep_alpha)
    lf0 = vuv = None
    if lf0_params is not None:
        lf0, vuv = ops.lf0_vuv(f0, f_off, lf0_params[0], lf0_params[1])
    if amplitude and sp is not None:
        ops.sqrt_inplace(sp)
    if want_ap or want_bap:
        main.wait_stream(side)
        for t in (x, f0):
            t.record_stream(side)
    f0 = f0.cpu().numpy()
    lf0 = lf0.cpu().numpy() if lf0 is not None else None
    vuv = vuv.cpu().numpy() if vuv is not None else None
    sp = sp.cpu().numpy() if sp is not None else None
    mc = mc.cpu().numpy() if mc is not None else None
    ap = ap.cpu().numpy() if ap is not None else None
    bap = bap.cpu().numpy() if bap is not None else None
    out = []
    for u in range(len(raws)):
        a, b = f_off[u], f_off[u + 1]
        out.append({"f0": f0[a:b],
                    "lf0": lf0[a:b, None] if lf0 is not None else None,
                    "vuv": vuv[a:b, None] if vuv is not None else None,
                    "sp": sp[a:b] if sp is not None else None,
                    "ap": ap[a:b] if ap is not None else None,
                    "mcep": mc[a:b] if mc is not None else None,
                    "bap": bap[a:b] if bap is not None else None})
    return out


class StreamStats(object):
    """Normalisation sums of the continuous streams (coded sp, lf0, bap) of a feature matrix,
    accumulated on the device over the batches of a gen_data run (fp64, fixed summation order):
    what MeanCovarianceExtractor / MeanStdDevExtractor.add_sample gather utterance by utterance
    (misc/normalisation/*.py).  `columns`: {stream name: (first column, width)}."""

    def __init__(self, columns, want_cov):
        self.columns = dict(columns)
        self.want_cov = bool(want_cov)
        self.count = 0
        self.sums = {}

    def add(self, cmp_dev):
        for name, (c0, w) in self.columns.items():
            if w == 0:
                continue
            acc = self.sums.get(name)
            if acc is None:
                self.sums[name] = ops.feature_stats(cmp_dev, c0, w, self.want_cov)
            else:
                ops.feature_stats(cmp_dev, c0, w, self.want_cov, sums=acc[0], second=acc[1])
        self.count += int(cmp_dev.shape[0])

    def store(self, name, extractor):
        """Hands the sums of one stream to an extractor (adds to what it already holds)."""
        if name not in self.sums:
            return
        first, second = (t.cpu().numpy() for t in self.sums[name])
        extractor.add_sums(self.count, first[None, :] if self.want_cov else first, second)


def extract_cmp_batch(raws, fs, hop_ms=5.0, n_fft=None, mcep_order=59, mcep_alpha=None,
                      f0_silence_threshold=30, lf0_zero=0, add_deltas=True, device=None,
                      mgc_gamma=None, f0_method="dio"):
    """wav(s) -> the `[T, 3*(ncs+1+nb)+1]` feature matrix of the reference's gen_data in one go,
    everything on the device: DIO + StoneMask, D4C -> coded bap, CheapTrick -> mcep, lf0 / V-UV
    with interpolate_lin, deltas and the stream layout (WorldFeatLabelGen.py:778-807, 809-889,
    1121-1172).  Returns (cmp [Ttot, W] f32 on the device, frame offsets [U+1])."""
    dev = _device(device)
    L = _lib.load()
    n_fft = n_fft or L.itts_cheaptrick_fft_size(int(fs), 71.0)
    if isinstance(raws, tuple):       # (samples of all utterances back to back, sample offsets)
        samples, x_off = raws
        x_off = [int(o) for o in x_off]
    else:
        x_off = offsets([len(r) for r in raws])
        samples = np.concatenate(raws) if len(raws) else np.empty(0)
    f_off = offsets([num_frames(b - a, fs, hop_ms) for a, b in zip(x_off[:-1], x_off[1:])])
    if isinstance(samples, torch.Tensor):     # float64 samples of gen_data's readers: page-locked (fetched asynchronously)
        x = samples if samples.is_cuda else samples.to(dev, non_blocking=True)     # or uploaded by the reader already
    else:
        x = torch.from_numpy(np.ascontiguousarray(samples, dtype=np.float64)).to(dev)
    f0 = estimate_f0(x, x_off, f_off, fs, hop_ms, f0_method)
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        _, bap = ops.d4c(x, x_off, f0, f_off, fs, hop_ms, n_fft, want_ap=False,
                         want_bap=torch.float32)
        lf0, vuv = ops.lf0_vuv(f0, f_off, f0_silence_threshold, lf0_zero)
    if mgc_gamma is None or mgc_gamma == 0.0:
        _, mc, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, hop_ms, n_fft, want_sp=False,
                                       order=mcep_order, alpha=mcep_alpha)
    else:       # sp_type "mgc": mel-generalized cepstrum of the CheapTrick envelope
        sp, _, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, hop_ms, n_fft, want_sp=True)
        mc = ops.mgcep(sp, mcep_order, mcep_alpha, mgc_gamma, input_is_power=True)
    main.wait_stream(side)
    for t in (x, f0, bap, lf0, vuv):
        t.record_stream(side)
        t.record_stream(main)
    return ops.assemble_cmp(mc, lf0, vuv, bap, f_off, add_deltas=add_deltas), f_off


def synthesise_features(f0, f_off, fs, n_fft, mc=None, alpha=None, sp=None, bap=None, ap=None, hop_ms=5.0,
                        preemphasis=0.0, dtype=torch.float32):
    """Device tensors in, waveform out: f0 [Ttot] f64 of utterances stored back to back, the envelope as mel-cepstra
    `mc` [Ttot, order + 1] f64 (with `alpha`) or as power spectra `sp`, the aperiodicity coded (`bap`) or decoded (`ap`).
    What has to be decoded first -- mc -> sp (WorldFeatLabelGen.py:925 through mgc2sp), bap -> ap (:940-941) -- runs on
    the side stream WHILE the synthesis works through everything it does on f0 alone (per-sample phase, pulse positions,
    noise): the two [Ttot, K] arrays are only needed by the pulse kernel.  Returns (y [Ytot], y_off)."""
    dev = f0.device
    main = torch.cuda.current_stream(dev)
    ready = ap_ready = None
    y_off = ops.synth_offsets(f_off, fs, hop_ms)      # (host work first: nothing between the two streams' launches)
    if mc is not None or ap is None:
        side = _side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            # (the envelope first -- the kernel of the unvoiced pulses waits for it alone --, the decode behind it with
            # an event of its own for the kernel of the voiced ones)
            if mc is not None:
                sp = ops.mgc2sp(mc, alpha, n_fft, want_pow=True)
            ready = torch.cuda.Event()
            ready.record(side)
            if ap is None:
                ap = ops.decode_aperiodicity(bap, fs, n_fft, voiced_f0=f0)      # (only the rows a voiced pulse reads)
                ap_ready = torch.cuda.Event()
                ap_ready.record(side)
        for t in (mc, bap, f0):
            if t is not None:
                t.record_stream(side)
    return ops.world_synthesize(f0, sp, ap, f_off, fs, hop_ms, preemphasis, dtype=dtype, spectra_ready=ready,
                                y_off=y_off, ap_ready=ap_ready)


def synthesise_batch(f0s, sps, baps, fs, n_fft, hop_ms=5.0, preemphasis=0.0, device=None,
                     out_dtype=np.float64, sp_is_amplitude=False):
    """f0s: list of [T] f64; sps: list of [T,K] f64 POWER spectra -- or, with `sp_is_amplitude`, amplitude spectra
    of any float type, widened and squared on the device after the upload (np.square(amp_sp, dtype=float64) of
    WorldFeatLabelGen.py:925: the same bits) --; baps: list of [T,nap] f64 coded
    aperiodicity. Returns list of waveforms (float32 samples; float64 container when
    out_dtype is float64, like scipy.signal.lfilter gives the reference)."""
    dev = _device(device)
    f_off = offsets([len(f) for f in f0s])
    f0 = torch.from_numpy(np.ascontiguousarray(np.concatenate(f0s), dtype=np.float64)).to(dev)
    if sp_is_amplitude:
        host = sps[0] if len(sps) == 1 else np.concatenate(sps)
        sp = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
        if sp.dtype != torch.float64:
            sp = sp.double()
        ops.square_inplace(sp)
    else:
        sp = torch.from_numpy(np.ascontiguousarray(np.concatenate(sps), dtype=np.float64)).to(dev)
    bap = torch.from_numpy(np.ascontiguousarray(np.concatenate(baps), dtype=np.float64)).to(dev)
    y, y_off = synthesise_features(f0, f_off, fs, n_fft, sp=sp, bap=bap, hop_ms=hop_ms, preemphasis=preemphasis,
                                   dtype=torch.float64 if out_dtype == np.float64 else torch.float32)
    y = y.cpu().numpy()
    return [y[y_off[u]:y_off[u + 1]] for u in range(len(f0s))]
