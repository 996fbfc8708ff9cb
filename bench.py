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


def cpu_baseline_ff(n_utts, max_seconds=20.0):
    """Reference stack (torch.nn.Linear/Tanh + MSELoss*mask + Adam) on the host cores, padded
    batch exactly like process_dataloader; bounded sample."""
    from idiaptts_amd.bench_support import (TorchRefFF, make_ff_batch, pad_batch, torch_ref_step)
    from idiaptts_amd.native_ff import FlatFFModel
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    dims, acts = (425, 512, 512, 187), ("tanh", "tanh", None)
    ref = TorchRefFF(FlatFFModel.reference_init(dims, 0), acts)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    x, y, lengths = make_ff_batch(n_utts, seed=0)
    lt = torch.from_numpy(lengths)
    xp, yp = pad_batch(x, lt), pad_batch(y, lt)
    torch_ref_step(ref, opt, xp, yp, lt)  # warm-up
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--utts-per-gpu", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node {}".format(args.gpus))

    from idiaptts_amd import lib
    lib.require_gpu()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=dev)

    from idiaptts_amd.bench_support import make_ff_batch
    from idiaptts_amd.native_ff import FlatFFModel, flops_per_frame

    dims, acts = (425, 512, 512, 187), ("tanh", "tanh", None)
    model = FlatFFModel(dims, acts, device=dev, seed=0)
    # weak scaling: every rank owns its own 32 utterances of the global batch
    n_batches = 4
    batches = []
    for b in range(n_batches):
        x, y, lengths = make_ff_batch(args.utts_per_gpu, seed=1000 * b + rank, device=dev)
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

    out = None
    if rank == 0:
        # dominant kernel roofline: the fp32-MFMA GEMM launches of one step, timed live with
        # events on the launch stream around a GEMM-only replay of the step's 8 GEMM launches.
        from idiaptts_amd import ops
        x, y, valid, nloc = batches[0]
        hs = model.forward(x)
        _, dz3 = ops.masked_mse(hs[-1], y, valid, float(nloc))
        stream = torch.cuda.current_stream()

        def gemms():
            h1 = ops.linear_fwd(x, model.weight(0), model.bias(0), 1)
            h2 = ops.linear_fwd(h1, model.weight(1), model.bias(1), 1)
            ops.linear_fwd(h2, model.weight(2), model.bias(2), 0)
            ops.linear_bwd_weight(dz3, h2, dw=model.weight(2, model.grads), want_bias=False)
            dz2 = ops.linear_bwd_input(dz3, model.weight(2), yprev=h2, act_prev=1)
            ops.linear_bwd_weight(dz2, h1, dw=model.weight(1, model.grads), want_bias=False)
            dz1 = ops.linear_bwd_input(dz2, model.weight(1), yprev=h1, act_prev=1)
            ops.linear_bwd_weight(dz1, x, dw=model.weight(0, model.grads), want_bias=False)

        gemms()
        torch.cuda.synchronize()
        ms = hip_event_time_ms(gemms, stream, 10)
        flops = flops_per_frame(dims) * nloc
        achieved = flops / (ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "gemm_f32_kernel (8 launches per step)",
                    "achieved": achieved, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_MFMA_F32_TFLOPS, "traffic": None,
                    "gemm_ms_per_step": ms}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline_ff(args.utts_per_gpu)
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
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
