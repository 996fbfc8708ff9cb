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
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


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
    return float(np.median(times))


def cpu_baseline_ff(n_utts, max_seconds=20.0):
    """Reference stack (torch.nn.Linear/Tanh + MSELoss*mask + Adam) on the host cores, padded
    batch exactly like process_dataloader; bounded sample."""
    from idiaptts_amd.bench_support import (TorchRefFF, make_ff_batch, pad_batch, torch_ref_step)
    from idiaptts_amd.native_ff import FlatFFModel
    ncpu = os.cpu_count() or 1
    dims, acts = (425, 512, 512, 187), ("tanh", "tanh", None)
    ref = TorchRefFF(FlatFFModel.reference_init(dims, 0), acts)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    x, y, lengths = make_ff_batch(n_utts, seed=0)
    lt = torch.from_numpy(lengths)
    xp, yp = pad_batch(x, lt), pad_batch(y, lt)
    # torch's intra-op pool over-subscribes badly with all hardware threads of a big host:
    # take the best of a short sweep (one step each) as the baseline's thread count.
    best, cores = None, ncpu
    for nt in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 32), min(ncpu, 16)}):
        torch.set_num_threads(nt)
        torch_ref_step(ref, opt, xp, yp, lt)  # warm-up for this pool size
        t = time.perf_counter()
        torch_ref_step(ref, opt, xp, yp, lt)
        t = time.perf_counter() - t
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
            "kind": "port",
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

    def analysis():
        f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs, hop), f_off, fs, hop)
        # D4C first, like world.analyse_batch: the mcep Newton loop reads trip counts back from
        # the device, so whatever is queued behind it starts late
        _, bap = ops.d4c(x, x_off, f0, f_off, fs, hop, n_fft, want_ap=False,
                         want_bap=torch.float32)
        _, mc, it = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, hop, n_fft, want_sp=False,
                                        order=order, alpha=alpha, want_iters=True)
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
    mc64 = mc.double()
    bap64 = bap.double()
    f0s = f0.clone()

    def synthesis():
        pw = ops.mgc2sp(mc64, alpha, n_fft, want_pow=True)
        apd = ops.decode_aperiodicity(bap64, fs, n_fft)
        return ops.world_synthesize(f0s, pw, apd, f_off, fs, hop)

    synthesis()
    sync()
    ms_sy = over_ranks(hip_event_median_ms(synthesis, stream, 5), dist.ReduceOp.MAX)
    frames = int(over_ranks(f_off[-1], dist.ReduceOp.SUM))
    audio_s = over_ranks(audio_s, dist.ReduceOp.SUM)
    res[key] = {
        "fs": fs, "utterances": n_utts * n_ranks, "n_gpus": n_ranks, "audio_seconds": audio_s,
        "frames": frames, "timing": "median of 5 passes, HIP events on the launch stream",
        "analysis_ms": ms_an, "analysis_rtf": ms_an * 1e-3 / audio_s,
        "analysis_frames_per_s": frames / (ms_an * 1e-3),
        "synthesis_ms": ms_sy, "synthesis_rtf": ms_sy * 1e-3 / audio_s,
        "mcep_newton_iters_mean": float(iters.float().mean().item()),
        # algorithmic HBM bytes per frame: fused analysis->features 640 + 248;
        # synthesis 8536
        "analysis_algorithmic_GBps": frames * (fs // 200 * 8 + (61 + L.itts_num_aperiodicities(fs)) * 4)
        / (ms_an * 1e-3) / 1e9,
        "synthesis_algorithmic_GBps": frames * ((n_fft // 2 + 1) * 16 + 8 + fs // 200 * 4)
        / (ms_sy * 1e-3) / 1e9,
    }
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
        ml_frames *= n_ranks                      # same lengths on every rank
        res["mlpg"] = {"utterances": 256 * n_ranks, "frames": ml_frames, "ms": ms_ml,
                       "frames_per_s": ml_frames / (ms_ml * 1e-3),
                       "algorithmic_GBps": ml_frames * 2000 / (ms_ml * 1e-3) / 1e9,
                       "frac_of_hbm_peak": ml_frames * 2000 / (ms_ml * 1e-3) / 1e9 / PEAK_HBM_GBS
                       / n_ranks}
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
        res[key]["cpu_baseline"] = {
            "kind": "port", "cores": 1,
            "sample": "{} utterances ({:.1f} s of audio) through the C oracle, one at a time on "
                      "one core".format(k, done_s),
            "analysis_rtf": t_an / done_s, "synthesis_rtf": t_sy / done_s}
    return res


def bilstm_section(dev, n_utts=64, steps=3, cell="LSTM", rank=0, world=1):
    """BASELINE config 3: 425 -> 3 x 512 BiLSTM -> 187, batch 64 padded utterances per GPU, Adam,
    fp32, through the drop-in module stack (RNNDyn + NamedLoss + fused HIP Adam).  With world > 1
    every rank trains on its own 64 utterances and the handler sums the frame-weighted gradients
    over RCCL (weak scaling); the reported rate is the whole job's."""
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

    h.process_batch(data, lens, 0, training=True)      # warm-up
    barrier()
    t0 = time.perf_counter()
    for s in range(steps):
        ld, _ = h.process_batch(data, lens, s + 1, training=True)
    barrier()
    dt = (time.perf_counter() - t0) / steps
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    frames = int(parallel.global_sum(int(lengths.sum()), device=dev))
    return {"bi" + cell.lower(): {
        "model": "425 -> 3x512 Bi{} -> 187".format(cell), "utterances_per_gpu": n_utts,
        "n_gpus": world, "valid_frames": frames, "max_frames": int(lengths.max()),
        "ms_per_step": dt * 1e3, "valid_frames_per_s": frames / dt,
        "loss": ld["MSELoss_acoustic_features"]}}


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
    t0 = time.perf_counter()
    epoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"resident_epoch": {"utterances": n_utts, "frames": n, "batch_utts": batch_utts,
                               "shard_GB": (x.numel() + y.numel()) * 4 / 1e9,
                               "epoch_ms": dt * 1e3, "valid_frames_per_s": n / dt}}


def spawn_ranks(args):
    """Launches `torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child
    process and returns its exit code (rank 0 of the child job prints the JSON line).  Nothing
    here initialises HIP: torch.cuda.device_count() only counts devices on this image."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--world-utts", type=int, default=256,
                    help="utterances in the WORLD feature-path section (0 = skip)")
    ap.add_argument("--world-fs", type=int, default=16000)
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional check only: all ranks on device 0, collectives over gloo "
                         "(never a measurement; the JSON line says so)")
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

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
    if world > 1:
        dist.all_reduce(counts)
    global_counts = counts.cpu().tolist()

    def step(i):
        x, y, valid, _ = batches[i % n_batches]
        return model.train_step(x, y, valid, global_counts[i % n_batches], lr=1e-3,
                                world_size=world)

    for i in range(args.warmup):
        step(i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    frames = sum(global_counts[i % n_batches] for i in range(args.steps))
    value = frames / dt

    # config 3 (and its GRU variant): every rank takes part, rank 0 reports
    rnn_extra = {}
    if args.bilstm_utts > 0:
        for cell in ("LSTM", "GRU"):
            rnn_extra.update(bilstm_section(dev, args.bilstm_utts, cell=cell, rank=rank,
                                            world=world))

    # config 5 (WORLD analysis / synthesis real-time factors, MLPG): every rank takes part
    world_extra = {}
    if args.world_utts > 0:
        with_cpu = not args.no_cpu_baseline and world == 1    # CPU baseline: rank 0 at N = 1 only
        world_extra = world_section(dev, args.world_utts, args.world_fs, with_cpu=with_cpu,
                                    rank=rank, n_ranks=world)
        # config 5 also quotes 48 kHz (fft 2048, 5 aperiodicity bands): a smaller batch
        world_extra.update(world_section(dev, max(4, args.world_utts // 4), 48000, cpu_seconds=12.0,
                                         with_cpu=with_cpu, with_mlpg=False, key="world_48k",
                                         rank=rank, n_ranks=world))

    out = None
    if rank == 0:
        # dominant kernel roofline: the fp32-MFMA GEMM launches of one step, timed live with
        # events on the launch stream around a GEMM-only replay of the step's 8 GEMM launches.
        from idiaptts_amd import ops
        x, y, valid, nloc = batches[0]
        hs = model.forward(x)
        M = x.shape[0]
        buf = model._rows_buffer       # the step's own padded-pitch activation buffers
        _, dz3 = ops.masked_mse(hs[-1], y, valid, float(nloc), grad=buf("dz_out", M, dims[3]))
        stream = torch.cuda.current_stream()

        def gemms():
            W, G = model.weight_padded, model.grads
            h1 = ops.linear_fwd(x, W(0), model.bias(0), 1, out=buf("h0", M, dims[1]))
            h2 = ops.linear_fwd(h1, W(1), model.bias(1), 1, out=buf("h1", M, dims[2]))
            ops.linear_fwd(h2, W(2), model.bias(2), 0, out=buf("h2", M, dims[3]))
            ops.linear_bwd_weight(dz3, h2, dw=W(2, G), want_bias=False)
            dz2 = ops.linear_bwd_input(dz3, W(2), yprev=h2, act_prev=1, out=buf("dz0", M, dims[2]))
            ops.linear_bwd_weight(dz2, h1, dw=W(1, G), want_bias=False)
            dz1 = ops.linear_bwd_input(dz2, W(1), yprev=h1, act_prev=1, out=buf("dz1", M, dims[1]))
            ops.linear_bwd_weight(dz1, x, dw=W(0, G), want_bias=False)

        for _ in range(20):      # the RNN sections above leave the clocks low: ramp up first
            gemms()
        torch.cuda.synchronize()
        ms = hip_event_time_ms(gemms, stream, 50)
        flops = flops_per_frame(dims) * nloc
        achieved = flops / (ms * 1e-3) / 1e12
        # HBM bytes per GEMM launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # see profiles/r1h_gemm_traffic.json); not re-measured inside bench.py.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1h_gemm_traffic.json")
        if os.path.isfile(tpath) and args.utts_per_gpu == 32:
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        roofline = {"bound": "mfma", "kernel": "gemm_f32_kernel (8 launches per step)",
                    "achieved": achieved, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_MFMA_F32_TFLOPS, "traffic": traffic,
                    "algorithmic_flops_per_launch": flops / 8.0,
                    "gemm_ms_per_step": ms, "avg_launch_us": ms * 1e3 / 8.0}
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # CPU baseline: rank 0 at N = 1 only
            cpu = cpu_baseline_ff(args.utts_per_gpu)
        extra = dict(world_extra)
        extra.update(rnn_extra)
        if world == 1 and args.world_utts > 0:
            extra.update(resident_epoch_section(dev))
        out = {
            "metric": "acoustic frames/sec (train)", "value": value, "unit": "valid frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "FF acoustic model 425->512(tanh)->512(tanh)->187 train step "
                                   "(fwd + masked MSE + bwd + Adam), {} utterances/GPU/step of "
                                   "2-10 s at 5 ms frames, packed valid frames".format(
                                       args.utts_per_gpu),
                       "utts_per_gpu": args.utts_per_gpu, "parallelism": "dp{}".format(world)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if share_gpu:
            out["shared_gpu"] = ("functional check only: {} ranks on ONE device, collectives over "
                                 "gloo -- not a measurement".format(world))
        out.update(extra)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
