#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native IdiapTTS hot path.

Metric (BASELINE.json): acoustic valid frames/sec in training, FF 425->512->512->187 (config 2),
32 utterances per GPU per step, fp32, synthetic data.  One "step" = forward +
masked-MSE + backward + (all-reduce) + Adam on one mini-batch already resident in HBM.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def usable_cores():
    """Cores this process may actually run on: the scheduler affinity mask cut down by the cgroup
    CPU quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us).  os.cpu_count() is the machine's
    logical CPU count whatever the container was given and is reported beside it, never used."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _oracle_utterance(args):
    """Pool worker: one synthetic utterance through the C oracle (analysis, then synthesis)."""
    fs, seed, seconds = args
    from idiaptts_amd.synthetic_audio import make_audio
    from oracle import capi
    from idiaptts_amd import lib
    L = lib.load()
    order, alpha = 59, L.itts_mcep_alpha(fs)
    n_fft = L.itts_cheaptrick_fft_size(fs, 71.0)
    r = make_audio(fs, seconds, seed)
    a = time.perf_counter()
    f0c, spc, apc = capi.wav2world(r, fs)
    bapc = capi.code_aperiodicity(apc, fs)
    mcc = capi.mcep(np.sqrt(spc), order, alpha)
    b = time.perf_counter()
    la = capi.mgc2sp_logamp(mcc, alpha, n_fft)
    pw = np.exp(la.astype(np.float32)).astype(np.float64) ** 2
    apd = capi.decode_aperiodicity(bapc, fs, n_fft)
    capi.synthesize(f0c, pw, apd, fs)
    c = time.perf_counter()
    return len(r) / fs, b - a, c - b


def _pool_warm(_):
    from oracle import capi   # load the oracle library in the worker before the clock starts
    capi.num_frames(16000, 16000)
    return os.getpid()


def cpu_baseline_world_pool(fs=16000, seconds=3.0, budget_s=40.0, single_core_analysis_rtf=None):
    """the reference's feature extraction is an embarrassingly parallel loop
    over files (WorldFeatLabelGen.py:996); its best case on this host is a pool of single-threaded
    worker processes.  The pool sizes are swept SMALLEST FIRST (1, 2, 4, ... up to the usable
    cores: affinity mask and cgroup quota, not os.cpu_count()) and every point is recorded; a point
    is only started while the budget lasts (its cost is predicted from the point before it, which
    had the same work per worker).  Workers are started and warmed first, then every worker gets
    one utterance of the same length (C oracle: analysis + synthesis) and the pool's wall time over
    the pool's audio is taken.  Runs in an interpreter that never loads torch or touches HIP."""
    import multiprocessing as mp
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[var] = "1"
    n = usable_cores()
    sizes, p = [], 1
    while p < n:
        sizes.append(p)
        p *= 4
    sizes = sorted(set(sizes + [max(1, n // 2), n]))
    rng = np.random.default_rng(5)
    sweep, best = [], None
    t_start = time.perf_counter()
    last_cost = 0.0
    for p in sizes:
        left = budget_s - (time.perf_counter() - t_start)
        if sweep and left < 1.3 * last_cost:
            break
        t_point = time.perf_counter()
        jobs = [(fs, 7000 + i, float(seconds)) for i in range(p)]
        with mp.get_context("fork").Pool(p) as pool:
            pool.map(_pool_warm, range(p), chunksize=1)
            t0 = time.perf_counter()
            res = pool.map(_oracle_utterance, jobs, chunksize=1)
            wall = time.perf_counter() - t0
        last_cost = time.perf_counter() - t_point
        audio = sum(r[0] for r in res)
        row = {"processes": p, "utterances": len(jobs), "audio_s": audio, "wall_s": wall,
               "analysis_plus_synthesis_rtf": wall / audio,
               "per_core_analysis_rtf": float(np.mean([r[1] / r[0] for r in res])),
               "per_core_synthesis_rtf": float(np.mean([r[2] / r[0] for r in res]))}
        sweep.append(row)
        if best is None or row["analysis_plus_synthesis_rtf"] < best["analysis_plus_synthesis_rtf"]:
            best = row
    one = single_core_analysis_rtf or sweep[0]["per_core_analysis_rtf"]
    degr = best["per_core_analysis_rtf"] / one
    out = {"kind": "port", "cores": best["processes"], "processes": best["processes"],
           "usable_cores": n, "logical_cpus": os.cpu_count(),
           "sample": "best of a sweep over the pool size (smallest first, {} points: {}): {} "
                     "single-threaded worker processes, {} utterances ({:.0f} s of audio, one per "
                     "worker), C oracle analysis + synthesis, workers started and warmed before the "
                     "clock".format(len(sweep), [r["processes"] for r in sweep], best["processes"],
                                    best["utterances"], best["audio_s"]),
           "analysis_plus_synthesis_rtf": best["analysis_plus_synthesis_rtf"],
           "per_core_analysis_rtf": best["per_core_analysis_rtf"],
           "per_core_synthesis_rtf": best["per_core_synthesis_rtf"],
           "per_worker_slowdown_vs_one_core": degr,
           "sweep_complete": len(sweep) == len(sizes),
           "sweep": [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()} for r in sweep]}
    if degr > 1.5:
        out["warning"] = ("a worker of the best pool ran {:.1f}x slower than the same code alone on one "
                          "core: the pool is limited by shared resources of the host (memory bandwidth, "
                          "SMT siblings, or fewer real cores than the affinity mask shows), not by the "
                          "code".format(degr))
    return out


if __name__ == "__main__" and "--cpu-pool-worker" in sys.argv:
    # child of the bench: the one-process-per-core C-oracle baseline, in an interpreter that never
    # loads torch or touches HIP (forking 256 workers out of the benchmark process itself, before
    # its GPU sections, slowed the launch-bound BiGRU section by 10 %)
    _a = sys.argv[sys.argv.index("--cpu-pool-worker") + 1:]
    print(json.dumps(cpu_baseline_world_pool(
        int(_a[0]), budget_s=float(_a[1]) if len(_a) > 1 else 40.0,
        single_core_analysis_rtf=float(_a[2]) if len(_a) > 2 and float(_a[2]) > 0 else None)))
    sys.exit(0)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F64_TFLOPS = 78.6         # MI355X_MICROARCH.md: fp64 vector = fp64 matrix peak
PEAK_HBM_GBS = 8000.0


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"cpu_model": model, "usable_cores": usable_cores(), "logical_cpus": os.cpu_count()}


def cpu_baseline_bilstm(max_seconds=25.0, threads=None, n_utts=16):
    """The reference's config-3 stack on the host: torch.nn.LSTM(425, 512, 3, bidirectional) on a
    PackedSequence (rnn_dyn/RNNWrapper.py:45-107) + Linear(1024, 187), masked MSE mean_per_frame,
    Adam.  Bounded sample: ONE training step on the first `n_utts` (16) utterances of config 3's batch of
    64, each cut to its first `cap` frames -- the host's cost is one set of small GEMMs per time step and
    layer whatever the batch width, so a step is linear in the padded length; `cap` is chosen from a
    pilot step so that the timed step fits the budget.  The extrapolation to the configuration (64
    utterances, 1 977 time steps) is stated beside the sample's own rate and is an UPPER bound for the
    host (it prices the 64-row step at the 16-row step's cost per time step)."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from idiaptts_amd.bench_support import make_ff_batch, pad_batch
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(425, 512, 3, bidirectional=True)
    fc = torch.nn.Linear(1024, 187)
    opt = torch.optim.Adam(list(lstm.parameters()) + list(fc.parameters()), lr=1e-3)
    torch.set_num_threads(threads or min(usable_cores(), 32))
    x, y, lengths = make_ff_batch(64, seed=7)       # the batch of the GPU section
    offs = np.concatenate([[0], np.cumsum(lengths)])
    full_T, full_frames = int(lengths.max()), int(lengths.sum())

    def step(n, cap):
        lt = torch.from_numpy(np.minimum(lengths[:n], cap))
        xs = torch.cat([x[offs[i]:offs[i] + int(lt[i])] for i in range(n)])
        ys = torch.cat([y[offs[i]:offs[i] + int(lt[i])] for i in range(n)])
        xp, yp = pad_batch(xs, lt).transpose(0, 1).contiguous(), pad_batch(ys, lt).transpose(0, 1).contiguous()
        T = xp.shape[0]
        mask = (torch.arange(T)[:, None] < lt[None, :]).unsqueeze(-1).float()
        t = time.perf_counter()
        out, _ = lstm(pack_padded_sequence(xp, lt, enforce_sorted=False))
        out, _ = pad_packed_sequence(out, total_length=T)
        pred = fc(out)
        loss = ((pred - yp) ** 2 * mask).sum() / (float(lt.sum()) * 187)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return time.perf_counter() - t, int(lt.sum()), T

    # The cost per time step grows with the padded length (autograd keeps every step's tensors, the host's caches
    # give out), so the length is doubled from 32 frames until the NEXT doubling would not fit what is left of the
    # budget at twice the last cost per time step; the last completed step is the sample.
    t_start = time.perf_counter()
    step(n_utts, 16)                             # thread pool, allocations, oneDNN primitives of this width
    cap = 32
    dt, frames, T = step(n_utts, cap)
    while cap < full_T:
        left = max_seconds - (time.perf_counter() - t_start)
        nxt = min(2 * cap, full_T)
        if 2.0 * (dt / T) * nxt > left:
            break
        cap = nxt
        dt, frames, T = step(n_utts, cap)
    per_time_step = dt / T
    full_step_s = per_time_step * full_T
    return {"kind": "port", "cores": torch.get_num_threads(), "value": frames / dt,
            "unit": "valid frames/s", "utterances": n_utts, "time_steps": T,
            "seconds_per_time_step": per_time_step,
            "extrapolated_config3_step_s": full_step_s,
            "extrapolated_config3_valid_frames_per_s_upper_bound": full_frames / full_step_s,
            "sample": "1 training step of torch.nn.LSTM(425,512,3,bidirectional)+Linear on the first {} of "
                      "config 3's 64 padded utterances, each cut to its first {} frames ({} valid frames, "
                      "{:.1f} s, {:.1f} ms per time step).  Extrapolation: the configuration's {} time steps "
                      "at this cost per time step = {:.0f} s per step of {} valid frames, i.e. at most {:.0f} "
                      "valid frames/s for the 64-utterance batch (an upper bound for the host: a 64-row step "
                      "costs at least what a 16-row step does)".format(
                          n_utts, cap, frames, dt, per_time_step * 1e3, full_T, full_step_s, full_frames,
                          full_frames / full_step_s)}


def hip_event_time_ms(fn, stream, iters):
    """Average duration of fn() measured with events recorded on `stream` (the stream the
    kernels are launched on)."""
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        start.record(stream)
        for _ in range(iters):
            fn()
        end.record(stream)
    end.synchronize()
    return start.elapsed_time(end) / iters


def committed_traffic(section):
    """HBM bytes per pass / step of a secondary section from the committed PMC passes of this round
    (profiles/r5_section_traffic.json: scripts/section_traffic.sh); None when the file is absent.
    Not re-measured inside bench.py (the counters need their own profiler runs)."""
    path = os.path.join(ROOT, "profiles", "r5_section_traffic.json")
    if not os.path.isfile(path):
        path = os.path.join(ROOT, "profiles", "r4_section_traffic.json")
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        row = json.load(f).get("sections", {}).get(section)
    return row["hbm_bytes"] if row else None


def committed_fractions(kernel_substring):
    """VALU-issue / LDS / wait fractions of a WORLD kernel from the committed SQ counter passes
    (profiles/r5_world_pmc_fractions.json: scripts/pmc_fractions.py); None when absent."""
    path = os.path.join(ROOT, "profiles", "r5_world_pmc_fractions.json")
    if not os.path.isfile(path):
        path = os.path.join(ROOT, "profiles", "r4_world_pmc_fractions.json")
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        rows = json.load(f)
    for k, v in rows.items():
        if kernel_substring in k:
            return dict(v, kernel=k)
    return None


def scratch_pool_stats(L):
    """reserved / used bytes and release threshold of the stream-ordered scratch pool: a section
    that ran with a pool below its working set shows up here (and 5x slower)"""
    import ctypes
    r, u, k = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    if L.itts_scratch_pool_stats(ctypes.byref(r), ctypes.byref(u), ctypes.byref(k)) != 0:
        return None
    return {"reserved_GB": r.value / 2 ** 30, "used_GB": u.value / 2 ** 30,
            "keep_threshold_GB": k.value / 2 ** 30}


def hip_event_median_ms(fn, stream, iters):
    """Median duration of fn() over `iters` passes, each bracketed by its own event pair on
    `stream` (the secondary sections: one slow pass -- a pool growing, a clock ramp -- must not
    decide the figure)."""
    times = []
    with torch.cuda.stream(stream):
        for _ in range(iters):
            start = torch.cuda.Event(enable_timing=True)
            end = torch.cuda.Event(enable_timing=True)
            start.record(stream)
            fn()
            end.record(stream)
            end.synchronize()
            times.append(start.elapsed_time(end))
    hip_event_median_ms.last = [float(t) for t in times]   # all passes, for the sections that report them
    return float(np.median(times))


def hip_event_queued_ms(fn, stream, calls, iters):
    """Median over `iters` passes of (one event pair around `calls` back-to-back fn()) / calls: a launch's
    duration as the stream sees it, the host side of a call hidden behind the launch before it."""
    times = []
    with torch.cuda.stream(stream):
        for _ in range(iters):
            start = torch.cuda.Event(enable_timing=True)
            end = torch.cuda.Event(enable_timing=True)
            start.record(stream)
            for _ in range(calls):
                fn()
            end.record(stream)
            end.synchronize()
            times.append(start.elapsed_time(end) / calls)
    return float(np.median(times))


def cpu_baseline_ff(n_utts, max_seconds=20.0):
    """Reference stack (torch.nn.Linear/Tanh + MSELoss*mask + Adam) on the host cores, padded
    batch exactly like process_dataloader; bounded sample."""
    from idiaptts_amd.bench_support import (TorchRefFF, make_ff_batch, pad_batch, torch_ref_step)
    from idiaptts_amd.native_ff import FlatFFModel
    ncpu = usable_cores()
    dims, acts = (425, 512, 512, 187), ("tanh", "tanh", None)
    ref = TorchRefFF(FlatFFModel.reference_init(dims, 0), acts)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    x, y, lengths = make_ff_batch(n_utts, seed=0)
    lt = torch.from_numpy(lengths)
    xp, yp = pad_batch(x, lt), pad_batch(y, lt)
    # torch's intra-op pool over-subscribes badly with all hardware threads of a big host:
    # take the best of a short sweep (one step each) as the baseline's thread count.
    best, cores, sweep = None, ncpu, []
    for nt in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 64), min(ncpu, 32),
                      min(ncpu, 16), min(ncpu, 8)}):
        torch.set_num_threads(nt)
        torch_ref_step(ref, opt, xp, yp, lt)  # warm-up for this pool size
        t = time.perf_counter()
        torch_ref_step(ref, opt, xp, yp, lt)
        t = time.perf_counter() - t
        sweep.append({"threads": nt, "step_s": round(t, 4)})
        if best is None or t < best:
            best, cores = t, nt
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    steps = 0
    while True:
        torch_ref_step(ref, opt, xp, yp, lt)
        steps += 1
        if time.perf_counter() - t0 > max_seconds or steps >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": float(lengths.sum()) * steps / dt, "unit": "valid frames/s", "cores": cores,
            "kind": "port", "thread_sweep": sweep,
            "sample": "{} training steps of the torch-CPU reference stack (nn.Linear/Tanh, masked "
                      "MSE mean_per_frame, Adam) on one {}-utterance padded batch ({} valid "
                      "frames)".format(steps, n_utts, int(lengths.sum()))}


def world_section(dev, n_utts, fs, cpu_seconds=25.0, with_cpu=True, with_mlpg=True, key="world",
                  rank=0, n_ranks=1):
    """WORLD feature path on `n_utts` synthetic utterances (inputs resident in HBM, GPU time by
    events on the launch stream): analysis wav -> (f0, mcep60, bap), synthesis
    (mcep60, bap, f0) -> wav, MLPG on the 187-dim cmp, and the C-oracle CPU baseline on a
    bounded sample (one utterance at a time on one core, like WorldFeatLabelGen.py:996).
    With n_ranks > 1 every rank owns its own `n_utts` utterances (utterances are independent: no
    data-path collective); times are the max over ranks after a barrier, audio and frames the
    sum, so the real-time factors are whole-job figures."""
    from idiaptts_amd import lib, ops, world
    from idiaptts_amd.bench_support import make_audio_batch
    L = lib.load()
    raws = make_audio_batch(n_utts, fs, seed=rank)
    hop = 5.0
    order, alpha = 59, L.itts_mcep_alpha(fs)
    n_fft = L.itts_cheaptrick_fft_size(fs, 71.0)
    x_off = world.offsets([len(r) for r in raws])
    f_off = world.offsets([world.num_frames(len(r), fs, hop) for r in raws])
    audio_s = x_off[-1] / fs
    x = torch.from_numpy(np.concatenate(raws)).to(dev)
    stream = torch.cuda.current_stream()
    res = {}

    overlap = os.environ.get("ITTS_BENCH_D4C_SIDE", "1") != "0"
    side = world._side_stream(dev)

    def analysis():
        f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs, hop), f_off, fs, hop)
        # as world.analyse_batch / extract_cmp_batch do it (the path gen_data takes): D4C on the
        # side stream beside CheapTrick + mcep -- they only share their inputs, and the mcep Newton
        # loop reads trip counts back from the device between launches
        if overlap:
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                _, bap = ops.d4c(x, x_off, f0, f_off, fs, hop, n_fft, want_ap=False,
                                 want_bap=torch.float32)
        else:
            _, bap = ops.d4c(x, x_off, f0, f_off, fs, hop, n_fft, want_ap=False,
                             want_bap=torch.float32)
        _, mc, it = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, hop, n_fft, want_sp=False,
                                        order=order, alpha=alpha, want_iters=True)
        if overlap:
            stream.wait_stream(side)
        return f0, mc, bap, it

    def over_ranks(value, op):
        if n_ranks == 1:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=op)
        return t.item()

    def sync():
        torch.cuda.synchronize()
        if n_ranks > 1:
            dist.barrier()

    f0, mc, bap, iters = analysis()
    sync()
    ms_an = over_ranks(hip_event_median_ms(analysis, stream, 5), dist.ReduceOp.MAX)
    an_passes = list(hip_event_median_ms.last)
    mc64 = mc.double()
    bap64 = bap.double()
    f0s = f0.clone()

    def synthesis():
        # as the product does it (world.synthesise_features): mel-cepstrum -> power spectrum and coded -> decoded
        # aperiodicity on the side stream while the synthesis works through what it does on f0 alone
        if os.environ.get("ITTS_BENCH_SYNTH_SIDE", "1") == "0":
            pw = ops.mgc2sp(mc64, alpha, n_fft, want_pow=True)
            apd = ops.decode_aperiodicity(bap64, fs, n_fft)
            return ops.world_synthesize(f0s, pw, apd, f_off, fs, hop)
        return world.synthesise_features(f0s, f_off, fs, n_fft, mc=mc64, alpha=alpha, bap=bap64, hop_ms=hop)

    synthesis()
    sync()
    ms_sy = over_ranks(hip_event_median_ms(synthesis, stream, 5), dist.ReduceOp.MAX)
    harvest = None
    if key == "world" and n_ranks == 1:
        # the alternative F0 estimator (pyworld.harvest): its own 1 ms grid, so it is timed on a
        # quarter of the utterances
        nh = max(1, n_utts // 4)
        xh, xoh, foh = x[:x_off[nh]], x_off[:nh + 1], f_off[:nh + 1]
        ops.harvest(xh, xoh, foh, fs, hop)
        sync()
        ms_h = hip_event_median_ms(lambda: ops.harvest(xh, xoh, foh, fs, hop), stream, 3)
        harvest = {"utterances": nh, "audio_seconds": xoh[-1] / fs, "ms": ms_h,
                   "rtf": ms_h * 1e-3 / (xoh[-1] / fs),
                   "note": "pyworld.harvest instead of dio + stonemask; HIP events, median of 3"}
    frames = int(over_ranks(f_off[-1], dist.ReduceOp.SUM))
    audio_s = over_ranks(audio_s, dist.ReduceOp.SUM)
    res[key] = {
        "fs": fs, "utterances": n_utts * n_ranks, "n_gpus": n_ranks, "audio_seconds": audio_s,
        "frames": frames, "timing": "median of 5 passes, HIP events on the launch stream",
        "analysis_ms": ms_an, "analysis_ms_passes": an_passes, "analysis_rtf": ms_an * 1e-3 / audio_s,
        "analysis_frames_per_s": frames / (ms_an * 1e-3),
        "synthesis_ms": ms_sy, "synthesis_rtf": ms_sy * 1e-3 / audio_s,
        "mcep_newton_iters_mean": float(iters.float().mean().item()),
        "harvest_f0": harvest,
        "scratch_pool": scratch_pool_stats(L),
        # algorithmic HBM bytes per frame: fused analysis->features 640 + 248;
        # synthesis 8536
        "analysis_algorithmic_GBps": frames * (fs // 200 * 8 + (61 + L.itts_num_aperiodicities(fs)) * 4)
        / (ms_an * 1e-3) / 1e9,
        "synthesis_algorithmic_GBps": frames * ((n_fft // 2 + 1) * 16 + 8 + fs // 200 * 4)
        / (ms_sy * 1e-3) / 1e9,
    }
    # rooflines (the WORLD kernels are HBM-bound by contract; in fact they are
    # fp64 FFT / LDS work, so the fp64 rate is stated beside the algorithmic bandwidth).  FLOPs per
    # frame: analysis ~ DIO 3.5 kFLOP/sample + CheapTrick / D4C FFTs + mcep Newton (measured
    # iterations x (2 FFT + 3 warping products + 60^3/3 solve)); synthesis 7 FFT-n_fft per pulse.
    w = res[key]
    it = w["mcep_newton_iters_mean"]
    lg = np.log2(n_fft)
    fft = 2.5 * n_fft * lg                     # real FFT of n_fft points
    K = n_fft // 2 + 1
    an_flops = (fs // 200) * 3500 + 3 * fft + 8 * 2.5 * 2 * n_fft * (lg + 1) + \
        2 * K * 60 + it * (2 * fft + 2 * K * (60 + 60 + 119) + 60 ** 3 / 3)
    sy_flops = 7 * fft * 1.3 + 2 * 60 * K
    nap = L.itts_num_aperiodicities(fs) if n_ranks >= 1 else 1
    w["analysis_roofline"] = {
        "bound": "hbm", "kernel": "mcls_solve_dpp_kernel + d4c_kernel + mcls_fused3_kernel (see profiles/)",
        "achieved": w["analysis_algorithmic_GBps"] / n_ranks, "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": w["analysis_algorithmic_GBps"] / n_ranks / PEAK_HBM_GBS,
        "traffic": committed_traffic("analysis_%d" % fs) if n_utts == (256 if fs <= 24000 else 64) else None,
        "what_binds_it": "fp64 VALU issue and LDS latency, not HBM: see issue_fractions (share of the chip's VALU "
                         "issue slots / LDS cycles in use, share of a wave's life spent waiting; SQ counters, "
                         "profiles/r5_world_pmc_fractions.json)",
        "issue_fractions": {k: committed_fractions(k) for k in ("mcls_solve_dpp", "d4c_kernel", "cheaptrick_wave",
                                                                 "mcls_fused3", "mcls_init_fused")} if fs <= 24000 else None,
        "algorithmic_bytes_per_frame": fs // 200 * 8 + (61 + nap) * 4,
        "fp64_tflops": frames * an_flops / (ms_an * 1e-3) / 1e12 / n_ranks,
        "fp64_frac_of_peak": frames * an_flops / (ms_an * 1e-3) / 1e12 / n_ranks / PEAK_F64_TFLOPS,
        "fp64_flops_per_frame_estimate": an_flops}
    w["synthesis_roofline"] = {
        "bound": "hbm", "kernel": "syn_pulse_wave_kernel" if n_fft == 1024 else "syn_pulse_kernel",
        "achieved": w["synthesis_algorithmic_GBps"] / n_ranks, "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": w["synthesis_algorithmic_GBps"] / n_ranks / PEAK_HBM_GBS,
        "traffic": committed_traffic("synthesis_%d" % fs) if n_utts == (256 if fs <= 24000 else 64) else None,
        "issue_fractions": {k: committed_fractions(k) for k in ("syn_pulse_wave", "gemm_f64_kernel<true, false, true")} if fs <= 24000 else None,
        "algorithmic_bytes_per_frame": (n_fft // 2 + 1) * 16 + 8 + fs // 200 * 4,
        "fp64_tflops": frames * sy_flops / (ms_sy * 1e-3) / 1e12 / n_ranks,
        "fp64_frac_of_peak": frames * sy_flops / (ms_sy * 1e-3) / 1e12 / n_ranks / PEAK_F64_TFLOPS}
    if with_mlpg:
        # MLPG on [T, 187] (62 static dims in 3 streams), 256 utterances of 2-10 s (SURVEY.md section 8d,
        # config 4): algorithmic 2000 B / frame
        from idiaptts_amd.bench_support import utterance_lengths
        ml_off = world.offsets(utterance_lengths(256, seed=5).tolist())
        ml_frames = ml_off[-1]
        feat = torch.randn(ml_frames, 186, dtype=torch.float64, device=dev)
        var = torch.rand(186, dtype=torch.float64, device=dev) * 0.99 + 0.01
        ops.mlpg_generation(feat, var, 62, ml_off)
        sync()
        ms_ml = over_ranks(hip_event_median_ms(lambda: ops.mlpg_generation(feat, var, 62, ml_off),
                                               stream, 7), dist.ReduceOp.MAX)
        # the launch itself: 8 calls queued back to back, events around them on the launch stream (a lone call holds
        # 40-50 us of host side -- allocations, sort, table, launch -- between its events, the device idle)
        ms_ml_q = over_ranks(hip_event_queued_ms(lambda: ops.mlpg_generation(feat, var, 62, ml_off), stream, 8, 5),
                             dist.ReduceOp.MAX)
        # the lone call again with a prepared plan (ops.MlpgPlan: offsets checked and sorted once, table in page-locked
        # memory, scratch kept): what a caller solving the streams of one batch pays from the second stream on
        ml_plan = ops.MlpgPlan(ml_off)
        ml_out = torch.empty((ml_frames, 62), dtype=torch.float64, device=dev)
        ops.mlpg_generation(feat, var, 62, ml_off, out=ml_out, plan=ml_plan)
        sync()
        ms_ml_p = over_ranks(hip_event_median_ms(lambda: ops.mlpg_generation(feat, var, 62, ml_off, out=ml_out,
                                                                             plan=ml_plan), stream, 7),
                             dist.ReduceOp.MAX)
        sync()
        ml_plan.close()
        del ml_out
        ml_frames *= n_ranks                      # same lengths on every rank
        gbs = ml_frames * 2000 / (ms_ml * 1e-3) / 1e9
        gbs_q = ml_frames * 2000 / (ms_ml_q * 1e-3) / 1e9
        # HBM bytes of one solve from the committed PMC passes (profiles/r5_section_traffic.json:
        # FETCH_SIZE x 2 + WRITE_SIZE on this same workload); not re-measured inside bench.py
        ml_traffic = committed_traffic("mlpg")
        res["mlpg"] = {"utterances": 256 * n_ranks, "frames": ml_frames, "ms": ms_ml,
                       "frames_per_s": ml_frames / (ms_ml * 1e-3),
                       "algorithmic_GBps": gbs,
                       "frac_of_hbm_peak": gbs / PEAK_HBM_GBS / n_ranks,
                       "ms_planned": ms_ml_p,
                       "frac_of_hbm_peak_planned": ml_frames * 2000 / (ms_ml_p * 1e-3) / 1e9 / PEAK_HBM_GBS / n_ranks,
                       "ms_queued": ms_ml_q,
                       "roofline": {"bound": "hbm", "kernel": "mlpg_ring_kernel (one launch)",
                                    "from": "8 calls queued back to back, HIP events around them on the launch stream, / 8 "
                                            "(`ms`, `frac_of_hbm_peak`: one call alone between its events, host side included)",
                                    "achieved": gbs_q / n_ranks, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": gbs_q / PEAK_HBM_GBS / n_ranks, "traffic": ml_traffic,
                                    "avg_launch_us": ms_ml_q * 1e3,
                                    "algorithmic_bytes_per_launch": ml_frames / n_ranks * 2000,
                                    "algorithmic_bytes_per_frame": 2000}}
        # the same batch with float32 rows (the acoustic model's output type; widened in the solve's loads):
        # 744 + 496 algorithmic bytes per frame
        feat32 = feat.float()
        ops.mlpg_generation(feat32, var, 62, ml_off)
        sync()
        ms_32 = over_ranks(hip_event_median_ms(lambda: ops.mlpg_generation(feat32, var, 62, ml_off),
                                               stream, 7), dist.ReduceOp.MAX)
        res["mlpg"]["float32_rows"] = {"ms": ms_32, "algorithmic_bytes_per_frame": 1240,
                                       "algorithmic_GBps": ml_frames * 1240 / (ms_32 * 1e-3) / 1e9}
        del feat32
        # the same solve at larger batches
        curve = []
        for n_u in (1024, 4096):
            off_b = world.offsets(utterance_lengths(n_u, seed=5).tolist())
            fr = off_b[-1]
            feat_b = torch.randn(fr, 186, dtype=torch.float64, device=dev)
            ops.mlpg_generation(feat_b, var, 62, off_b)
            sync()
            ms_b = over_ranks(hip_event_median_ms(lambda: ops.mlpg_generation(feat_b, var, 62, off_b),
                                                  stream, 5), dist.ReduceOp.MAX)
            ms_bq = over_ranks(hip_event_queued_ms(lambda: ops.mlpg_generation(feat_b, var, 62, off_b), stream, 4, 3),
                               dist.ReduceOp.MAX)
            curve.append({"utterances": n_u, "frames": fr, "ms": ms_b, "ms_queued": ms_bq,
                          "algorithmic_GBps": fr * 2000 / (ms_b * 1e-3) / 1e9,
                          "frac_of_hbm_peak": fr * 2000 / (ms_b * 1e-3) / 1e9 / PEAK_HBM_GBS,
                          "frac_of_hbm_peak_queued": fr * 2000 / (ms_bq * 1e-3) / 1e9 / PEAK_HBM_GBS})
            del feat_b
        res["mlpg"]["batch_curve"] = curve
    if with_cpu:
        from oracle import capi
        t0 = time.perf_counter()
        done_s = 0.0
        t_an = t_sy = 0.0
        k = 0
        while time.perf_counter() - t0 < cpu_seconds and k < len(raws):
            r = raws[k]
            a = time.perf_counter()
            f0c, spc, apc = capi.wav2world(r, fs)
            bapc = capi.code_aperiodicity(apc, fs)
            mcc = capi.mcep(np.sqrt(spc), order, alpha)
            b = time.perf_counter()
            la = capi.mgc2sp_logamp(mcc, alpha, n_fft)
            pw = np.exp(la.astype(np.float32)).astype(np.float64) ** 2
            apd = capi.decode_aperiodicity(bapc, fs, n_fft)
            capi.synthesize(f0c, pw, apd, fs)
            c = time.perf_counter()
            t_an += b - a
            t_sy += c - b
            done_s += len(r) / fs
            k += 1
        if res[key].get("harvest_f0"):
            # the C restatement of Harvest on the first utterance (about 0.5 s of CPU)
            a = time.perf_counter()
            capi.harvest(raws[0], fs)
            res[key]["harvest_f0"]["cpu_oracle_rtf"] = (time.perf_counter() - a) / (len(raws[0]) / fs)
        res[key]["cpu_baseline"] = {
            "kind": "port", "cores": 1,
            "sample": "{} utterances ({:.1f} s of audio) through the C oracle, one at a time on "
                      "one core".format(k, done_s),
            "analysis_rtf": t_an / done_s, "synthesis_rtf": t_sy / done_s}
    return res


def rnn_flops_per_frame(in_dim=425, H=512, layers=3, out_dim=187, gates=4):
    """fwd + bwd GEMM FLOPs per valid frame of the (bi)recurrent stack: input projections, the
    recurrent products (h W_hh^T forward, dG W_hh and dW_hh backward), dX of every layer but the
    first, and the output layer."""
    f = 0
    for layer in range(layers):
        k = in_dim if layer == 0 else 2 * H
        per_dir = 2 * (k + H) * gates * H                    # forward
        per_dir += 2 * k * gates * H + 2 * 2 * H * gates * H    # dW_ih, dG W_hh, dW_hh
        if layer > 0:
            per_dir += 2 * k * gates * H                     # dX
        f += 2 * per_dir
    return f + 3 * 2 * 2 * H * out_dim


def bilstm_section(dev, n_utts=64, steps=6, cell="LSTM", rank=0, world=1, key=None):
    """BASELINE config 3: 425 -> 3 x 512 BiLSTM -> 187, `n_utts` padded utterances per GPU, Adam,
    fp32, through the drop-in module stack (RNNDyn + NamedLoss + fused HIP Adam).  With world > 1
    every rank trains on its own utterances and the handler sums the frame-weighted gradients over
    RCCL; the reported rate is the whole job's.  Timed with events on the launch stream over
    `steps` steps after two warm-up steps."""
    import types
    from idiaptts_amd import parallel
    from idiaptts_amd.bench_support import make_ff_batch
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler
    from idiaptts_amd.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    from idiaptts_amd.src.neural_networks.pytorch.models import rnn_dyn
    from idiaptts_amd.src.neural_networks.pytorch.models.NamedForwardWrapper import \
        NamedForwardWrapper
    torch.manual_seed(0)
    hp = types.SimpleNamespace(model_type="RNNDYN-3_Bi{}_512-1_FC_187".format(cell),
                               batch_first=False, dropout=0.0)
    h = Handler()
    h.create_model(NamedForwardWrapper.Config(rnn_dyn.convert_legacy_to_config((425,), hp),
                                              input_names=["questions"], batch_first=False,
                                              name="AcousticModel",
                                              output_names=["pred_acoustic_features"]))
    h.set_optimiser("Adam", lr=1e-3)
    h.set_losses([NamedLoss.Config(name="MSELoss_acoustic_features", type_="MSELoss",
                                   seq_mask="acoustic_features_mask",
                                   input_names=["acoustic_features", "pred_acoustic_features"],
                                   batch_first=False)])
    x, y, lengths = make_ff_batch(n_utts, seed=7 + 100 * rank)
    offs = np.concatenate([[0], np.cumsum(lengths)])
    batch = [{"questions": x[offs[i]:offs[i + 1]], "acoustic_features": y[offs[i]:offs[i + 1]]}
             for i in range(n_utts)]
    data, lens = Handler.prepare_batch(batch, batch_first=False, mask_keys=("acoustic_features",))
    data = {k: v.to(dev) for k, v in data.items()}

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for s in range(2):
        ld, _ = h.process_batch(data, lens, s, training=True)      # warm-up
    barrier()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for s in range(steps):
        # blocking=False: the NaN guard of a step is looked at when the next one has been queued
        ld, _ = h.process_batch(data, lens, s + 2, training=True, blocking=False)
    e1.record(stream)
    e1.synchronize()
    h.finish_batches()
    ld = {k: float(v) for k, v in ld.items()}
    dt = e0.elapsed_time(e1) * 1e-3 / steps
    barrier()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    frames = int(parallel.global_sum(int(lengths.sum()), device=dev))
    gates = 4 if cell == "LSTM" else 3
    tflops = rnn_flops_per_frame(gates=gates) * frames / dt / 1e12
    return {key or ("bi" + cell.lower()): {
        "model": "425 -> 3x512 Bi{} -> 187".format(cell), "utterances_per_gpu": n_utts,
        "utterances_global": n_utts * world, "n_gpus": world, "valid_frames": frames,
        "max_frames": int(lengths.max()), "steps_timed": steps,
        "timing": "HIP events on the launch stream, 2 warm-up steps",
        "ms_per_step": dt * 1e3, "valid_frames_per_s": frames / dt,
        "loss": ld["MSELoss_acoustic_features"],
        "roofline": {"bound": "mfma", "kernel": "gemm_ring_kernel + rnn_persist_fwd_kernel<{0}> + rnn_persist_bwd_kernel<{0}>".format(
                         gates),
                     "achieved": tflops / world, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                     "frac": tflops / world / PEAK_MFMA_F32_TFLOPS,
                     "traffic": committed_traffic("bi" + cell.lower()) if n_utts == 64 else None,
                     "algorithmic_flops_per_frame": rnn_flops_per_frame(gates=gates)}}}


def duration_mlpg_section(dev, n_utts=256):
    """BASELINE config 4: duration model (per-phone question vector 416 -> 2 x 512 tanh -> 5 state
    durations) + MLPG of the 187-dim acoustic trajectory, inference only: utterances per second
    through [duration forward, duration post-processing, MLPG] with inputs resident."""
    from idiaptts_amd import ops, world
    from idiaptts_amd.bench_support import utterance_lengths
    from idiaptts_amd.native_ff import FlatFFModel
    lengths = utterance_lengths(n_utts, seed=5)
    phones = np.maximum(lengths // 18, 1)                 # ~90 ms per phone at 5 ms frames
    P = int(phones.sum())
    model = FlatFFModel((416, 512, 512, 5), ("tanh", "tanh", None), device=dev, seed=2)
    g = torch.Generator(device=dev).manual_seed(9)
    q = (torch.rand((P, 416), generator=g, device=dev) < 0.05).float()
    ml_off = world.offsets(lengths.tolist())
    feat = torch.randn(ml_off[-1], 186, dtype=torch.float64, device=dev)
    var = torch.rand(186, dtype=torch.float64, device=dev) * 0.99 + 0.01
    stream = torch.cuda.current_stream()

    def run():
        dur = model.forward(q)[-1]
        frames = torch.clamp(torch.round(dur), min=0).to(torch.int64)     # DurationModelTrainer.forward
        out = ops.mlpg_generation(feat, var, 62, ml_off)
        return frames, out

    run()
    torch.cuda.synchronize()
    ms = hip_event_median_ms(run, stream, 7)
    ms_dur = hip_event_median_ms(lambda: model.forward(q), stream, 7)
    return {"duration_mlpg": {
        "utterances": n_utts, "phones": P, "frames": int(ml_off[-1]), "ms": ms,
        "duration_model_ms": ms_dur, "utterances_per_s": n_utts / (ms * 1e-3),
        "phones_per_s": P / (ms_dur * 1e-3), "frames_per_s": ml_off[-1] / (ms * 1e-3),
        "timing": "median of 7 passes, HIP events on the launch stream"}}


def api_single_call_section(dev, fs=16000, seconds=6.6, repeats=20, with_cpu=True):
    """ONE utterance through the reference's one-utterance signatures, numpy in -> numpy out, host <-> device copies
    and every host-side step included (median wall clock of `repeats` calls after two warm-up calls):
      analysis   WorldFeatLabelGen.world_extract_features (:778-807) + AudioProcessing.extract_mcep (:142-153),
                 what WorldFeatLabelGen.extract_features (:809-889, called once per file by gen_data :996) does
      synthesis  WorldFeatLabelGen.world_features_to_raw (:909-945)
      mlpg       MLPG().generation (misc/mlpg.py:94-127) of the 60 mcep trajectories (one stream, as the reference
                 calls it per stream)
    beside the same three steps of the C oracle on one core.  Every other WORLD figure of this file is a 64-512
    utterance batch; this is what a caller who changes nothing pays per call."""
    from idiaptts_amd import lib
    from idiaptts_amd.misc.mlpg import MLPG
    from idiaptts_amd.src.data_preparation.audio.AudioProcessing import AudioProcessing
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    from idiaptts_amd.synthetic_audio import make_audio
    L = lib.load()
    alpha = L.itts_mcep_alpha(fs)
    raw = make_audio(fs, seconds, 31)
    rng = np.random.default_rng(2)

    def analysis():
        amp_sp, lf0, vuv, bap = WorldFeatLabelGen.world_extract_features(raw, fs, 5)
        return amp_sp, lf0, vuv, bap, AudioProcessing.extract_mcep(amp_sp, 60, alpha)

    amp_sp, lf0, vuv, bap, mcep = analysis()
    T = len(lf0)
    feats = rng.standard_normal((T, 180))
    cov = np.diag(rng.uniform(0.01, 1.0, 180))

    def synthesis():
        return WorldFeatLabelGen.world_features_to_raw(amp_sp, lf0.copy(), vuv.copy(), bap, fs)

    mlpg = MLPG()

    def smooth():
        return mlpg.generation(feats, cov, 60)

    def median_ms(fn):
        for _ in range(2):
            fn()
        times = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        return float(np.median(times)) * 1e3, float(np.min(times)) * 1e3

    out = {"fs": fs, "audio_seconds": seconds, "frames": int(T), "repeats": repeats,
           "timing": "median (and minimum) wall clock of one call, numpy in -> numpy out"}
    for key, fn in (("analysis", analysis), ("synthesis", synthesis), ("mlpg", smooth)):
        med, best = median_ms(fn)
        out[key] = {"ms": med, "ms_min": best}
    out["analysis"]["rtf"] = out["analysis"]["ms"] * 1e-3 / seconds
    out["synthesis"]["rtf"] = out["synthesis"]["ms"] * 1e-3 / seconds
    if with_cpu:
        from oracle import capi
        n_fft = L.itts_cheaptrick_fft_size(fs, 71.0)
        t0 = time.perf_counter()
        f0c, spc, apc = capi.wav2world(raw, fs)
        bapc = capi.code_aperiodicity(apc, fs)
        mcc = capi.mcep(np.sqrt(spc), 59, alpha)
        t1 = time.perf_counter()
        pw = np.exp(capi.mgc2sp_logamp(mcc, alpha, n_fft).astype(np.float32)).astype(np.float64) ** 2
        capi.synthesize(f0c, pw, capi.decode_aperiodicity(bapc, fs, n_fft), fs)
        t2 = time.perf_counter()
        capi.mlpg(feats, np.diag(cov).copy(), 60)
        t3 = time.perf_counter()
        out["cpu_oracle_one_core_ms"] = {"analysis": (t1 - t0) * 1e3, "synthesis": (t2 - t1) * 1e3,
                                         "mlpg": (t3 - t2) * 1e3}
    return {"api_single_call": out}


def gen_data_section(n_utts=512, batch_utts=64):
    """The drop-in WorldFeatLabelGen.gen_data end to end: wav files ->
    per-stream .npz archives with deltas + normalisation statistics, file I/O, host <-> device copies and
    all host work included (median of 5 passes into an empty output directory, all passes listed; two more passes
    over the existing archives beside them).  The files live on /dev/shm where it
    is writable (the boxes' local disks throttle write-back after a few hundred MB: the same run took 0.16 s
    on one box and 0.71 s on another), else in the default temporary directory; `dir` in the row says which."""
    import tempfile
    from scipy.io import wavfile
    from idiaptts_amd.bench_support import make_audio_batch
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=base) as tmp:
        wav_dir, out_dir = os.path.join(tmp, "wav"), os.path.join(tmp, "out")
        os.makedirs(wav_dir)
        ids = []
        audio = 0.0
        for i, xw in enumerate(make_audio_batch(n_utts, 16000, seed=0)):
            wavfile.write(os.path.join(wav_dir, "u%03d.wav" % i), 16000, (xw * 32767).astype(np.int16))
            ids.append("u%03d" % i)
            audio += len(xw) / 16000.0
        gen = WorldFeatLabelGen(out_dir, add_deltas=True, num_coded_sps=60, batch_utts=batch_utts)
        gen.gen_data(wav_dir, out_dir, "ids.txt", id_list=ids[:batch_utts])
        import shutil
        times, times_over = [], []
        for k in range(7):
            # passes 0-4 write into an EMPTY output directory, as a first extraction does (the directory of the pass
            # before is removed outside the clock); passes 5-6 write over the archives that are there -- every one of
            # them is then opened and its member list read first (an archive with foreign keys is merged, not replaced)
            if k < 5:
                shutil.rmtree(out_dir, ignore_errors=True)
            t0 = time.perf_counter()
            gen.gen_data(wav_dir, out_dir, "ids.txt", id_list=ids)
            (times if k < 5 else times_over).append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return {"gen_data": {"utterances": n_utts, "batch_utts": batch_utts, "audio_seconds": audio,
                         "seconds": dt, "rtf": dt / audio, "dir": base or tempfile.gettempdir(),
                         "passes_s": [round(t, 4) for t in times],
                         "passes_over_existing_archives_s": [round(t, 4) for t in times_over],
                         "what": "wav files -> mcep60 / lf0 / vuv / bap .npz with deltas + "
                                 "mean-covariance files, file I/O included"}}


def resident_epoch_section(dev, n_utts=1024, batch_utts=32):
    """one epoch of the FF model over an HBM-resident FrameShard (synthetic,
    LJSpeech-shaped utterances), mini-batches of `batch_utts` shuffled utterances gathered on the
    device as packed valid frames, flat train step.  The rate includes the index upload and the
    row gathers, i.e. everything the reference does per step between disk and optimiser."""
    from idiaptts_amd.bench_support import utterance_lengths
    from idiaptts_amd.native_ff import FlatFFModel
    from idiaptts_amd.src.data_preparation.FrameShard import FrameShard
    lengths = utterance_lengths(n_utts, seed=11)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    n = int(offsets[-1])
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.zeros((n, 428), dtype=torch.float32, device=dev)
    x[:, :425] = torch.rand((n, 425), generator=g, device=dev)
    y = torch.zeros((n, 188), dtype=torch.float32, device=dev)
    y[:, :187] = torch.randn((n, 187), generator=g, device=dev)
    shard = FrameShard(x, y, offsets, ["utt%05d" % i for i in range(n_utts)], 425, 187)
    model = FlatFFModel((425, 512, 512, 187), ("tanh", "tanh", None), device=dev, seed=0)
    order = torch.randperm(n_utts, generator=torch.Generator().manual_seed(5)).numpy()

    def epoch():
        for b in range(0, n_utts, batch_utts):
            xb, yb, lens = shard.gather(order[b:b + batch_utts])
            valid = torch.ones(xb.shape[0], dtype=torch.uint8, device=dev)
            model.train_step(xb, yb, valid, float(lens.sum()), lr=1e-3)

    epoch()
    torch.cuda.synchronize()
    times = []
    for _ in range(3):          # (the host drives 32 steps per ms of GPU work here: one descheduled moment doubles an epoch)
        t0 = time.perf_counter()
        epoch()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return {"resident_epoch": {"utterances": n_utts, "frames": n, "batch_utts": batch_utts,
                               "shard_GB": (x.numel() + y.numel()) * 4 / 1e9,
                               "epoch_ms": dt * 1e3, "epochs_ms": [round(t * 1e3, 2) for t in times],
                               "valid_frames_per_s": n / dt}}


def trainer_epoch_section(dev, n_utts=1024, batch_utts=32, n_val=16, with_bilstm=True):
    """One epoch of `AcousticModelTrainer.train` over `n_utts` synthetic LJSpeech-shaped utterances through
    the PUBLIC API (reference model_trainers/ModularTrainer.py:379-560, AcousticModelTrainer.py), files on
    disk in the legacy layout the reference's tests use (`<id>.questions` [T, 425] and `cmp_mcep60/<id>.cmp`
    [T, 187] raw float32 + min-max / mean-covariance `.bin` files): the module path --
    PyTorchDatareadersDataset + loader + collate + RNNDyn modules + NamedLoss + fused HIP Adam -- with and
    without the device batch cache, `hparams.resident_dataset` (flat feed-forward step), and the BiLSTM
    model on the cached module path.  Three epochs each; the THIRD epoch's `handler.train` call is the one
    reported (wall clock, device synchronised)."""
    import logging
    import shutil
    import tempfile
    from idiaptts_amd.bench_support import utterance_lengths
    from idiaptts_amd.src.model_trainers.AcousticModelTrainer import AcousticModelTrainer
    root = tempfile.mkdtemp(prefix="itts_trainer_epoch_")
    out = {"utterances": n_utts, "batch_utts": batch_utts, "model": "RNNDYN-2_TANH_512-1_FC_187"}
    try:
        lengths = utterance_lengths(n_utts + n_val, seed=11)
        ids = ["utt%05d" % i for i in range(n_utts + n_val)]
        qdir, wdir = os.path.join(root, "questions"), os.path.join(root, "WORLD")
        os.makedirs(qdir)
        os.makedirs(os.path.join(wdir, "cmp_mcep60"))
        rng = np.random.default_rng(17)
        t0 = time.perf_counter()
        for i, T in zip(ids, lengths):
            rng.random((int(T), 425), dtype=np.float32).tofile(os.path.join(qdir, i + ".questions"))
            rng.standard_normal((int(T), 187), dtype=np.float32).tofile(os.path.join(wdir, "cmp_mcep60", i + ".cmp"))
        np.stack([np.zeros(425), np.ones(425)]).astype(np.float64).tofile(os.path.join(qdir, "min-max.bin"))
        for stream, D in (("mcep60", 180), ("lf0", 3), ("bap", 3)):      # mean 0, covariance I: the data is N(0, 1)
            path = os.path.join(wdir, "cmp_mcep60", stream + "-mean-covariance.bin")
            with open(path, "wb") as f:
                np.array([0, D + 1], dtype=np.int32).tofile(f)
                np.concatenate([np.zeros((1, D)), np.eye(D)]).astype(np.float64).tofile(f)
        out["files_GB"] = float(lengths.sum()) * (425 + 187) * 4 / 1e9
        out["files_written_s"] = time.perf_counter() - t0
        logging.getLogger().setLevel(logging.WARNING)
        # Rows (all through AcousticModelTrainer.train, hparams.dataset_num_workers_gpu = 4 reader threads as the
        # reference's default of four workers):
        #   module_path               nothing opted into: the readers' rows stay in HBM after the first epoch
        #                             (hparams.dataset_device_cache, default on) and later batches are gathered there
        #   module_path_uncached      hparams.dataset_device_cache = False: every epoch reads, normalises, pads and
        #                             uploads again, as the reference does (what `module_path` was before round 6)
        #   resident_dataset          hparams.resident_dataset: FrameShards + the flat feed-forward step
        #   module_path_cached_bilstm BASELINE config 3's model (3 x 512 BiLSTM, 64 utterances a batch) on the cached
        #                             module path, beside the `bilstm` section's step on one fixed batch
        only = os.environ.get("ITTS_TRAINER_EPOCH_ONLY")          # (scripts/prof_trainer_epoch.py)
        rows = (("module_path", {}), ("module_path_uncached", {"cache": False}), ("resident_dataset", {"resident": True}),
                ("module_path_cached_bilstm", {"model": "RNNDYN-3_BiLSTM_512-1_FC_187", "batch": 64}))
        for key, opt in rows:
            if only and key != only:
                continue
            if key == "module_path_cached_bilstm" and not with_bilstm:
                continue
            workers = 4
            hp = AcousticModelTrainer.create_hparams()
            hp.num_questions = 425
            hp.voice = "full"
            hp.out_dir = os.path.join(root, key)
            hp.frame_size_ms = 5
            hp.num_coded_sps = 60
            hp.seed = 1
            hp.epochs = 3
            hp.use_gpu = True
            hp.dataset_num_workers_gpu = workers
            hp.model_type = opt.get("model", "RNNDYN-2_TANH_512-1_FC_187")
            hp.batch_size_train = opt.get("batch", batch_utts)
            hp.batch_size_val = n_val
            hp.val_set_perc = n_val / float(n_utts + n_val)
            hp.test_set_perc = 0.0
            hp.start_with_test = False
            hp.epochs_per_test = 1
            hp.epochs_per_checkpoint = 1000      # (no per-epoch checkpoints; the final model is saved once)
            hp.use_best_as_final_model = False
            hp.optimiser_args["lr"] = 0.001
            hp.model_name = "bench_model"
            hp.world_dir = wdir
            hp.resident_dataset = bool(opt.get("resident", False))
            hp.dataset_device_cache = bool(opt.get("cache", True))
            trainer = AcousticModelTrainer(**AcousticModelTrainer.legacy_support_init(wdir, qdir, ids, 425, hp))
            trainer.init(hp)
            handler = trainer.model_handler
            epochs = []
            inner = handler.train

            def timed_train(*a, _inner=inner, **kw):
                torch.cuda.synchronize()
                t = time.perf_counter()
                r = _inner(*a, **kw)
                torch.cuda.synchronize()
                epochs.append(time.perf_counter() - t)
                return r

            handler.train = timed_train
            t0 = time.perf_counter()
            trainer.train(hp)
            total = time.perf_counter() - t0
            frames = int(sum(lengths[ids.index(i)] for i in trainer.id_list_train))
            steps = -(-len(trainer.id_list_train) // hp.batch_size_train)
            out[key] = {"train_utterances": len(trainer.id_list_train), "train_frames": frames,
                        "dataloader_workers": workers, "model": hp.model_type, "batch_utts": hp.batch_size_train,
                        "epoch_s": epochs[-1], "epoch_s_all": [round(e, 4) for e in epochs],
                        "steps_per_epoch": steps, "ms_per_step": epochs[-1] * 1e3 / steps,
                        "valid_frames_per_s": frames / epochs[-1],
                        "train_call_s": total,
                        "what": "third epoch's handler.train (the first holds warm-up{}); train_call_s is the whole "
                                "trainer.train call: 3 epochs + 3 validation passes + final checkpoint".format(
                                    " and the one-off staging of the shards to HBM" if hp.resident_dataset else
                                    " and the one upload of every utterance's rows" if hp.dataset_device_cache else "")}
            loader = getattr(handler, "dataloader_train", None)
            if hasattr(loader, "stats"):
                out[key]["device_cache"] = dict(loader.stats, cached_GB=loader.cached_bytes() / 1e9)
            del trainer, handler
    except Exception as e:      # a bench row must not take the headline line down with it
        out["error"] = "{}: {}".format(type(e).__name__, e)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return {"trainer_epoch": out}


class SclkSampler:
    """Shader clock of GPU `index` while a section runs (sysfs pp_dpm_sclk, the level marked '*',
    sampled every 20 ms by a thread): boxes of the pool sustain different clocks under the fp32-MFMA
    load, and the roofline fraction is quoted against the 2.4 GHz peak."""

    def __init__(self, index=0, period_s=0.02):
        import glob
        self.period_s = period_s
        self.path = None
        self.samples = []
        self._stop = False
        self._thread = None
        try:       # the sysfs card of torch device `index`: match its PCI address
            props = torch.cuda.get_device_properties(index)
            addr = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
            for card in glob.glob("/sys/class/drm/card*/device"):
                if os.path.realpath(card).endswith(addr) and os.path.isfile(card + "/pp_dpm_sclk"):
                    self.path = card + "/pp_dpm_sclk"
        except (AttributeError, RuntimeError, OSError):
            pass

    def _read(self):
        try:
            with open(self.path) as f:
                for line in f:
                    if "*" in line:
                        return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError, TypeError):
            pass
        return None

    def __enter__(self):
        import threading
        if self.path is not None:
            def loop():
                while not self._stop:
                    v = self._read()
                    if v is not None:
                        self.samples.append(v)
                    time.sleep(self.period_s)
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join()

    def summary(self):
        if not self.samples:
            return None
        v = sorted(self.samples)
        return {"sclk_mhz_median": v[len(v) // 2], "sclk_mhz_min": v[0], "sclk_mhz_max": v[-1],
                "samples": len(v)}


def exchange_plan(n, ff_params, rnn_params, ff_step_ms, rnn_step_ms, rnn_small_step_ms=None):
    """What every section exchanges per step / call at n GPUs and what that is expected to cost, so that
    a SCALE record can be checked against it.  xGMI on this node is a full mesh of
    point-to-point links, 7 per GPU at ~153 GB/s each (MI355X guide): a RING all-reduce
    is bound by ONE link -- each rank moves 2 (n - 1) / n of the buffer over it -- whereas the direct form
    SURVEY section 5 prescribes for the large buffer (reduce-scatter + all-gather, every rank talking to all
    n - 1 peers at once) moves 2 S / n per link.  Both are given; RCCL picks its own algorithm, so a measured
    exchange should fall between them.  ~5 us of latency per hop (2 (n - 1) hops in a ring, 2 in the direct form)."""
    if n <= 1:
        return None
    link = 153e9

    def ring(bytes_):
        return 2.0 * (n - 1) / n * bytes_ / link + 2 * (n - 1) * 5e-6

    def direct(bytes_):
        return 2.0 * bytes_ / n / link + 2 * 5e-6

    ff_b, rnn_b = 4 * ff_params, 4 * rnn_params
    # FF: three per-layer all-reduces issued behind each layer's weight-gradient launch; only the last
    # (first layer's 0.87 MB) has nothing to hide behind
    first_layer_b = 4 * (428 * 512 + 512)
    stats_b = 8 * (1 + 187 + 187 * 187)
    plan = {
        "n_gpus": n, "xgmi_link_GBps": link / 1e9, "xgmi_links_per_gpu": 7, "assumed_hop_latency_us": 5.0,
        "ff_train_step": {"collective": "3 all-reduce(sum) of the flat fp32 gradient segments, asynchronous",
                          "bytes_per_rank_per_step": ff_b,
                          "all_reduce_ms": {"ring": ring(ff_b) * 1e3, "direct_rs_ag": direct(ff_b) * 1e3},
                          "predicted_exposed_ms": {"ring": ring(first_layer_b) * 1e3,
                                                   "direct_rs_ag": direct(first_layer_b) * 1e3},
                          "predicted_efficiency": {"ring": ff_step_ms / (ff_step_ms + ring(first_layer_b) * 1e3),
                                                   "direct_rs_ag": ff_step_ms / (ff_step_ms + direct(first_layer_b) * 1e3)}},
        "bilstm_bigru_step": {"collective": "bucketed all-reduce(sum) of the flat gradient arena, started per bucket "
                                            "while backward runs (HipAdam.begin_overlapped_allreduce); the last "
                                            "bucket (first layer, ~1/6 of the arena) has nothing to hide behind",
                              "bytes_per_rank_per_step": rnn_b,
                              "all_reduce_ms": {"ring": ring(rnn_b) * 1e3, "direct_rs_ag": direct(rnn_b) * 1e3},
                              "predicted_exposed_ms": {"ring": ring(rnn_b / 6) * 1e3,
                                                       "direct_rs_ag": direct(rnn_b / 6) * 1e3},
                              "predicted_efficiency": ({"ring": rnn_step_ms / (rnn_step_ms + ring(rnn_b / 6) * 1e3),
                                                        "direct_rs_ag": rnn_step_ms / (rnn_step_ms + direct(rnn_b / 6) * 1e3),
                                                        "nothing_overlapped_ring": rnn_step_ms / (rnn_step_ms + ring(rnn_b) * 1e3)}
                                                       if rnn_step_ms else None)},
        "world_analysis_synthesis_mlpg": {"collective": "none in the data path", "bytes_per_rank_per_step": 0,
                                          "predicted_efficiency": 1.0},
        "gen_data": {"collective": "1 all-reduce(sum) of the normalisation sums per call",
                     "bytes_per_rank_per_call": stats_b, "predicted_exposed_ms": ring(stats_b) * 1e3}}
    # BASELINE config 3 as worded -- "batch 64 padded utterances, 8 GPUs": two readings, both run by this file.
    # Per GPU the step is (recurrences: ~4 us per frame step whatever the batch) + (products: proportional to the
    # rows); with 8 of the 64 utterances per GPU the products shrink 8-fold, the recurrences do not.
    if rnn_step_ms:
        small = rnn_small_step_ms if rnn_small_step_ms else None
        exposed = ring(rnn_b / 6) * 1e3
        plan["config_3_as_worded"] = {
            "one_gpu_step_ms_64_utterances": rnn_step_ms,
            "measured_step_ms_8_utterances_per_gpu": small,
            "predicted_speedup_at_8_gpus": {
                # weak form: 8 x 64 utterances per step in the time of one step + the exposed exchange
                "64_utterances_per_gpu": 8.0 * rnn_step_ms / (rnn_step_ms + exposed),
                # strong form: the same 64 utterances, 8 per GPU; the step measured at 8 utterances (on one device
                # when this line comes from a shared-GPU run), else the recurrences' share of the 64-utterance step
                "64_utterances_in_all": rnn_step_ms / ((small if small else 0.55 * rnn_step_ms) + exposed)},
            "note": "the strong form cannot reach the 6 x BASELINE asks for: the recurrences' 48.9 of the step's 96.5 ms "
                    "(profiles/r5an_bilstm_step_trace.txt) do not shrink with the batch; the weak form can"}
    return plan


def visible_gpus():
    """Number of GPUs this process may use, WITHOUT touching HIP in this process: the compute nodes of
    the KFD topology (sysfs) that have SIMDs AND whose render node this user can open (a container or
    cgroup may show the topology of the whole machine but grant only some /dev/dri/renderD* nodes), cut
    down by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one of them is set.
    When sysfs gives no answer a child interpreter asks torch.cuda.device_count()."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
            node = "/dev/dri/renderD%d" % minor
            if minor >= 0 and os.path.exists("/dev/dri") and not os.access(node, os.R_OK | os.W_OK):
                continue
            n += 1
        except (OSError, ValueError):
            pass
    if n == 0:
        import subprocess
        try:
            res = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
            n = int(res.stdout.strip().splitlines()[-1])
        except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
            n = 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([t for t in v.split(",") if t.strip() != ""])
            n = min(n, listed) if n else listed
    return n


def spawn_ranks(args):
    """Launches `torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child
    process and returns its exit code (rank 0 of the child job prints the JSON line).  Nothing
    here initialises HIP (the devices are counted in sysfs), and the ranks are fresh interpreters
    started with subprocess, never forks of a process that has touched the GPU."""
    import socket
    import subprocess
    n_dev = visible_gpus()
    env = dict(os.environ)
    if n_dev < args.gpus and not (args.share_gpu or env.get("ITTS_BENCH_SHARE_GPU") == "1"):
        print("bench.py: --gpus {} but only {} device(s) visible (use --share-gpu for a functional "
              "check of the N > 1 path on one device)".format(args.gpus, n_dev), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--utts-per-gpu", type=int, default=32)
    ap.add_argument("--dump-steps", action="store_true", help="every timed step's duration in the JSON line")
    ap.add_argument("--ramp-steps", type=int, default=80,
                    help="untimed steps run before the --warmup steps so that the shader clock has "
                         "settled (about 1 ms each; 0 = none); reported as clock_ramp in the JSON line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=100.0,
                    help="wall-time budget of ALL host baselines together (FF stack, BiLSTM stack, "
                         "C oracle on one core, C oracle process pool): each gets a quarter")
    ap.add_argument("--world-utts", type=int, default=256,
                    help="utterances in the WORLD feature-path section (0 = skip)")
    ap.add_argument("--world-fs", type=int, default=16000)
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional check only: all ranks on device 0, collectives over gloo "
                         "(never a measurement; the JSON line says so)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: initialise RCCL with one rank and issue every collective of "
                         "the N > 1 path anyway (a one-rank sum is the identity)")
    ap.add_argument("--trainer-utts", type=int, default=1024,
                    help="utterances of the trainer_epoch row (AcousticModelTrainer.train through the public "
                         "API on synthetic files in a temporary directory, ~3 GB at 1024; 0 = skip)")
    ap.add_argument("--bilstm-utts", type=int, default=64,
                    help="utterances per GPU of the BiLSTM / BiGRU (config 3) section (0 = skip)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, before this process
        # makes any HIP call (a process that touched the GPU must never be replaced or forked)
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    if args.share_gpu:
        os.environ["ITTS_BENCH_SHARE_GPU"] = "1"

    cpu_extra = {}
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline

    from idiaptts_amd import lib
    lib.require_gpu()
    # Functional check of the N > 1 control flow on a one-GPU box: ITTS_BENCH_SHARE_GPU=1 puts all
    # ranks on device 0 and runs the collectives over gloo (RCCL refuses two ranks on one device).
    # Never set for measurements.
    share_gpu = os.environ.get("ITTS_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if args.force_dist:
            from idiaptts_amd import parallel
            parallel.force_collectives(True)

    from idiaptts_amd.bench_support import make_ff_batch
    from idiaptts_amd.native_ff import FlatFFModel, flops_per_frame

    dims, acts = (425, 512, 512, 187), ("tanh", "tanh", None)
    model = FlatFFModel(dims, acts, device=dev, seed=0)
    # weak scaling: every rank owns its own 32 utterances of the global batch
    n_batches = 4
    batches = []
    for b in range(n_batches):
        x, y, lengths = make_ff_batch(args.utts_per_gpu, seed=1000 * b + rank, device=dev)
        x = model.pack_input(x)   # collate-time layout: row pitch padded to 16 B
        valid = torch.ones(x.shape[0], dtype=torch.uint8, device=dev)
        batches.append((x, y, valid, int(lengths.sum())))
    # global valid-frame count per step (identical on all ranks)
    counts = torch.tensor([b[3] for b in batches], dtype=torch.float64, device=dev)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        dist.all_reduce(counts)
    global_counts = counts.cpu().tolist()

    def step(i):
        x, y, valid, _ = batches[i % n_batches]
        return model.train_step(x, y, valid, global_counts[i % n_batches], lr=1e-3,
                                world_size=world)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # Clock ramp, declared in the JSON line ("clock_ramp"): the chip takes tens of milliseconds of
    # load to reach the shader clock it then holds, and `--warmup 5`
    # is 5.6 ms of work.  args.ramp_steps untimed steps run first; the W
    # warm-up steps of the contract follow, then the barrier, then EXACTLY K timed steps.
    ramp_steps = int(args.ramp_steps)      # the same count on every rank (a step holds collectives)
    t_ramp = time.perf_counter()
    for i in range(ramp_steps):
        step(i)
    torch.cuda.synchronize()
    t_ramp = time.perf_counter() - t_ramp
    for i in range(args.warmup):
        step(i)

    # everything the timed region needs is created BEFORE the barrier
    stream = torch.cuda.current_stream()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    evs[0].record(stream)
    for i in range(args.steps):
        step(i)
        evs[i + 1].record(stream)
    t_launched = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    # GEMM-only replay of the six GEMM launches of a step, RIGHT BEHIND the timed region (same clocks, same
    # batches in the same order, events on the launch stream): how much of a step its dominant kernel is.
    # (Round 4 ran this replay after the RNN sections, on batch 0 only: its 1.022 ms per step exceeded the
    # 1.0095 ms step it was supposed to be a part of.)
    from idiaptts_amd import ops

    def gemms(i):
        x, y, valid, nloc = batches[i % n_batches]
        M = x.shape[0]
        buf = model._rows_buffer       # the step's own padded-pitch activation buffers
        W, G = model.weight_padded, model.grads
        h1 = ops.linear_fwd(x, W(0), model.bias(0), 1, out=buf("h0", M, dims[1]))
        h2 = ops.linear_fwd(h1, W(1), model.bias(1), 1, out=buf("h1", M, dims[2]))
        ops.linear_fwd(h2, W(2), model.bias(2), 0, out=buf("h2", M, dims[3]))
        # the backward launches of the step (native_ff.loss_and_backward): weight, bias and input
        # gradient of a layer share one launch; dz_out holds the last timed step's loss gradient
        dz3, dz2, dz1 = buf("dz_out", M, dims[3]), buf("dz0", M, dims[2]), buf("dz1", M, dims[1])
        ops.linear_bwd(dz3, h2, W(2), W(2, G), model.bias(2, G), dz2, yprev=h2, act_prev=1)
        ops.linear_bwd(dz2, h1, W(1), W(1, G), model.bias(1, G), dz1, yprev=h1, act_prev=1)
        ops.linear_bwd_weight(dz1, x, dw=W(0, G), db=model.bias(0, G))

    replay_ms = []
    for _ in range(3):
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        for i in range(args.steps):
            gemms(i)
        r1.record(stream)
        r1.synchronize()
        replay_ms.append(r0.elapsed_time(r1) / args.steps)
    # The shader clock is sampled AFTER the timed region, over 60 more steps of the same load.  Rounds 2
    # and 3 sampled it (sysfs pp_dpm_sclk, from a thread) DURING the timed steps, and that was the "fixed
    # 2 ms" of the driver's 20-step protocol: the first ~20 ms of such reads slow every step by 10-40 %
    # (per-step events, profiles/r4m_step_traces.txt: 1.13 ms per step with the sampler, 1.007 without,
    # same box, same binary) -- the instrument was perturbing the measurement.
    clock = SclkSampler(local_rank, period_s=0.002)
    with clock:
        for i in range(60):
            step(i)
        torch.cuda.synchronize()
    dt_events = evs[0].elapsed_time(evs[-1]) * 1e-3
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    if dist_on:
        tt = torch.tensor([dt, dt_events], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_events = tt.tolist()
    frames = sum(global_counts[i % n_batches] for i in range(args.steps))
    value = frames / dt

    # config 3 (and its GRU variant): every rank takes part, rank 0 reports
    rnn_extra = {}
    if args.bilstm_utts > 0:
        for cell in ("LSTM", "GRU"):
            rnn_extra.update(bilstm_section(dev, args.bilstm_utts, cell=cell, rank=rank,
                                            world=world))
        if world > 1 and args.bilstm_utts >= world:
            # SURVEY.md section 8(d) defines config 3 as 64 utterances GLOBAL (strong scaling: 64 / N
            # per GPU); "bilstm" above is 64 per GPU (weak scaling).  Both are reported.
            rnn_extra.update(bilstm_section(dev, args.bilstm_utts // world, cell="LSTM", rank=rank,
                                            world=world, key="bilstm_global_batch"))
            rnn_extra["bilstm_global_batch"]["scaling"] = "strong"
        if world == 1 and args.bilstm_utts >= 64:
            # what ONE GPU of eight would run per step under config 3's strong reading (64 utterances in all): the
            # exchange plan's prediction for that reading rests on this measured step
            rnn_extra.update(bilstm_section(dev, args.bilstm_utts // 8, steps=4, cell="LSTM", key="bilstm_eighth_batch"))
        rnn_extra["bilstm"]["scaling"] = "weak"

    # config 5 (WORLD analysis / synthesis real-time factors, MLPG): every rank takes part
    world_extra = {}
    if args.world_utts > 0:
        with_cpu = not args.no_cpu_baseline and world == 1    # CPU baseline: rank 0 at N = 1 only
        world_extra = world_section(dev, args.world_utts, args.world_fs, with_cpu=with_cpu,
                                    cpu_seconds=min(25.0, args.cpu_budget_s / 6),
                                    rank=rank, n_ranks=world)
        # config 5 also quotes 48 kHz (fft 2048, 5 aperiodicity bands): a smaller batch
        world_extra.update(world_section(dev, max(4, args.world_utts // 4), 48000, cpu_seconds=min(12.0, args.cpu_budget_s / 12),
                                         with_cpu=with_cpu, with_mlpg=False, key="world_48k",
                                         rank=rank, n_ranks=world))

    out = None
    if rank == 0:
        # dominant kernel roofline FROM THE TIMED STEPS: algorithmic flops of the K timed batches (this rank's
        # frames) over the event time of the K timed steps on the launch stream -- the six GEMM launches plus
        # everything else a step holds (loss reduction, split-K slab sums, Adam), so `achieved` is a lower
        # bound of the rate the GEMM launches run at; `gemm_replay` is the GEMM-only replay of the same
        # batches queued right behind the timed region.
        local_frames = sum(batches[i % n_batches][3] for i in range(args.steps))
        flops = flops_per_frame(dims) * local_frames
        ms_step_events = dt_events / args.steps * 1e3
        achieved = flops / dt_events / 1e12
        gemm_ms = float(np.median(replay_ms))
        # HBM bytes per GEMM launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # see profiles/r4_gemm_traffic.json); not re-measured inside bench.py.
        traffic = None
        for name in ("r6_gemm_traffic.json", "r5_gemm_traffic.json", "r4_gemm_traffic.json", "r3g_gemm_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.isfile(tpath) and args.utts_per_gpu == 32:
                with open(tpath) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
                break
        n_launch = 6    # fwd1, fwd2, fwd3 + MSE, (dW3, db3, dX2), (dW2, db2, dX1), (dW1, db1)
        roofline = {"bound": "mfma",
                    "kernel": "gemm_ring_kernel / gemm_ring_pair_kernel (6 GEMM launches per step)",
                    "achieved": achieved, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_MFMA_F32_TFLOPS, "traffic": traffic,
                    "from": "the {} timed steps: {:.4g} GFLOP of GEMM work per step over {:.4f} ms per step "
                            "(HIP events on the launch stream around the timed region)".format(
                                args.steps, flops / args.steps / 1e9, ms_step_events),
                    "algorithmic_flops_per_launch": flops / args.steps / n_launch,
                    "avg_launch_us": ms_step_events * 1e3 / n_launch,
                    "gemm_replay": {
                        "what": "the six GEMM launches alone, same batches and order, queued right behind the "
                                "timed region; median of 3 passes of {} steps".format(args.steps),
                        "gemm_ms_per_step": gemm_ms, "passes_ms_per_step": [round(v, 4) for v in replay_ms],
                        "achieved": flops / args.steps / (gemm_ms * 1e-3) / 1e12,
                        "frac": flops / args.steps / (gemm_ms * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS,
                        "avg_launch_us": gemm_ms * 1e3 / n_launch,
                        "share_of_step": gemm_ms / ms_step_events,
                        "consistent": bool(gemm_ms <= ms_step_events)},
                    "shader_clock_under_the_same_load_after_the_timed_steps": clock.summary()}
        cpu = None
        if want_cpu:   # CPU baseline: rank 0 at N = 1 only
            cpu = cpu_baseline_ff(args.utts_per_gpu, max_seconds=min(20.0, args.cpu_budget_s / 4))
            cpu.update(host_info())
            if args.bilstm_utts > 0:
                rnn_extra["bilstm"]["cpu_baseline"] = cpu_baseline_bilstm(
                    max_seconds=min(30.0, args.cpu_budget_s / 4), threads=cpu["cores"])
        extra = dict(world_extra)
        extra.update(rnn_extra)
        if want_cpu and "world" in extra:
            # CPU baselines come last (rank 0 at N = 1 only); the process pool runs in a child
            # interpreter of its own
            import subprocess
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-pool-worker",
                                  str(args.world_fs), str(args.cpu_budget_s / 4),
                                  str(extra["world"].get("cpu_baseline", {}).get("analysis_rtf", 0.0))],
                                 stdout=subprocess.PIPE, text=True)
            lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
            if res.returncode == 0 and lines:
                extra["world"]["cpu_baseline_pool"] = json.loads(lines[-1])
        if world == 1 and args.world_utts > 0:
            extra.update(resident_epoch_section(dev))
            if args.trainer_utts > 0:
                extra.update(trainer_epoch_section(dev, args.trainer_utts, with_bilstm=args.bilstm_utts > 0))
                te = extra["trainer_epoch"]
                if "module_path_cached_bilstm" in te and "bilstm" in extra:
                    te["module_path_cached_bilstm"]["bilstm_section_ms_per_step"] = extra["bilstm"]["ms_per_step"]
                    te["module_path_cached_bilstm"]["bilstm_section_valid_frames_per_s"] = \
                        extra["bilstm"]["valid_frames_per_s"]
                    te["module_path_cached_bilstm"]["rate_vs_bilstm_section"] = \
                        te["module_path_cached_bilstm"]["valid_frames_per_s"] / extra["bilstm"]["valid_frames_per_s"]
                if "resident_epoch" in extra and "error" not in extra["trainer_epoch"]:
                    extra["trainer_epoch"]["bare_flat_step_valid_frames_per_s"] = value
                    extra["trainer_epoch"]["resident_epoch_section_valid_frames_per_s"] = \
                        extra["resident_epoch"]["valid_frames_per_s"]
            extra.update(duration_mlpg_section(dev))
            extra.update(api_single_call_section(dev, with_cpu=want_cpu))
            extra.update(gen_data_section(min(512, max(8, 2 * args.world_utts)),
                                          min(64, max(4, args.world_utts // 4))))
        out = {
            "metric": "acoustic frames/sec (train)", "value": value, "unit": "valid frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "effective_warmup": ramp_steps + args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "ms_per_step_hip_events": dt_events / args.steps * 1e3,
            "timed_region": {"wall_ms": dt * 1e3, "hip_events_ms": dt_events * 1e3,
                             "host_launch_ms": t_launched * 1e3,
                             "step_ms_first8": [round(v, 4) for v in step_ms[:8]],
                             "step_ms_median": float(np.median(step_ms)),
                             "step_ms_max": float(np.max(step_ms)),
                             **({"step_ms_all": [round(v, 4) for v in step_ms]} if args.dump_steps else {}),
                             "note": "value and ms_per_step are the wall clock (barrier + synchronize on "
                                     "both sides, max over ranks); the event pair brackets the same K "
                                     "steps on the launch stream"},
            "clock_ramp": {"untimed_steps_before_warmup": ramp_steps, "ramp_ms": t_ramp * 1e3,
                           "why": "the shader clock needs tens of ms of load to settle; the W warm-up "
                                  "steps of the contract run after it"},
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "FF acoustic model 425->512(tanh)->512(tanh)->187 train step "
                                   "(fwd + masked MSE + bwd + Adam), {} utterances/GPU/step of "
                                   "2-10 s at 5 ms frames, packed valid frames; weak scaling under --gpus N "
                                   "(every rank its own {} utterances; config 3 likewise: 64 per GPU)".format(
                                       args.utts_per_gpu, args.utts_per_gpu),
                       "utts_per_gpu": args.utts_per_gpu, "parallelism": "dp{}".format(world)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        rnn_params = 2 * (4 * 512 * (428 + 512) + 8 * 512) + 2 * 2 * (4 * 512 * (1024 + 512) + 8 * 512) + 1024 * 187 + 187
        out["exchange_plan"] = exchange_plan(
            world if world > 1 else 8, model.numel, rnn_params, dt / args.steps * 1e3,
            rnn_extra.get("bilstm", {}).get("ms_per_step"),
            rnn_extra.get("bilstm_global_batch", rnn_extra.get("bilstm_eighth_batch", {})).get("ms_per_step"))
        if world == 1:
            out["exchange_plan"]["note"] = "prediction for 8 GPUs from this run's single-GPU step times (nothing here has run on more than one GPU)"
        if share_gpu:
            out["shared_gpu"] = ("functional check only: {} ranks on ONE device, collectives over "
                                 "gloo -- not a measurement".format(world))
        if args.force_dist:
            out["forced_collectives"] = ("RCCL initialised with {} rank(s); every collective of the "
                                         "N > 1 path was issued".format(world))
        out.update(extra)
        print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
