"""Writes profiles/<tag>_{bench_line.json, ff_kernel_stats.csv, bench_kernel_stats.csv, summary.md}
from a gpurun_out/<tag>/ directory holding
    bench_line.json             python3 bench.py
    ff/ff_kernel_stats.csv      rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --world-utts 0 --bilstm-utts 0 --no-cpu-baseline
    full/full_kernel_stats.csv  rocprofv3 ... -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline
usage: python3 scripts/profile_summary.py <tag> [title]"""
import csv
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else "state of this round"
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, tag + "_bench_line.json"))
shutil.copy(os.path.join(src, "ff", "ff_kernel_stats.csv"), os.path.join(dst, tag + "_ff_kernel_stats.csv"))
shutil.copy(os.path.join(src, "full", "full_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
d = json.loads(open(os.path.join(dst, tag + "_bench_line.json")).read().strip().split("\n")[-1])
ff = list(csv.DictReader(open(os.path.join(dst, tag + "_ff_kernel_stats.csv"))))
full = list(csv.DictReader(open(os.path.join(dst, tag + "_bench_kernel_stats.csv"))))
g = [r for r in ff if "gemm_f32_kernel" in r["Name"]]
gt, gc = sum(float(r["TotalDurationNs"]) for r in g), sum(int(r["Calls"]) for r in g)
w, w48, ml, bl, bg, re_ = d["world"], d["world_48k"], d["mlpg"], d["bilstm"], d["bigru"], d["resident_epoch"]
rf = d["roofline"]
o = ["# Profile set {}: {}\n".format(tag, title),
     "`python3 bench.py` (defaults: 1 GPU, 200 steps, 20 warm-up) on an MI355X, line in `{}_bench_line.json`:\n".format(tag),
     "| item | value |\n|---|---|",
     "| FF 425-512-512-187 train step, 32 utterances / step | %.3f ms -> %.1f M valid frames/s |" % (d["ms_per_step"], d["value"] / 1e6),
     "| fp32-MFMA GEMMs of one step (8 launches, events on the launch stream, live in bench.py) | %.3f ms, avg %.1f us per launch, %.1f TFLOP/s = %.1f %% of 157.3 |" % (rf["gemm_ms_per_step"], rf["avg_launch_us"], rf["achieved"], 100 * rf["frac"]),
     "| the same launches in `rocprofv3 --kernel-trace --stats` (`%s_ff_kernel_stats.csv`: `python3 bench.py --world-utts 0 --bilstm-utts 0 --no-cpu-baseline`) | %d GEMM launches, avg %.1f us (kernel time only; the live figure includes the gaps between the launches) |" % (tag, gc, gt / gc / 1e3),
     "| HBM bytes per GEMM launch (PMC FETCH_SIZE x 2 + WRITE_SIZE, `%s_gemm_traffic.json`) | %.0f MB |" % (tag, rf["traffic"] / 1e6),
     "| CPU baseline (torch reference stack, %d threads) | %.0f valid frames/s |" % (d["cpu_baseline"]["cores"], d["cpu_baseline"]["value"]),
     "| WORLD analysis, %d utterances = %.0f s of 16 kHz audio | %.2f ms, RTF %.2e (C oracle, 1 core: %.3f) |" % (w["utterances"], w["audio_seconds"], w["analysis_ms"], w["analysis_rtf"], w["cpu_baseline"]["analysis_rtf"]),
     "| WORLD synthesis, same batch | %.2f ms, RTF %.2e (C oracle: %.3f) |" % (w["synthesis_ms"], w["synthesis_rtf"], w["cpu_baseline"]["synthesis_rtf"]),
     "| Harvest F0 (pyworld.harvest) on %d of those utterances = %.0f s | %.1f ms, RTF %.2e |" % ((w.get("harvest_f0") or {}).get("utterances", 0), (w.get("harvest_f0") or {}).get("audio_seconds", 0.0), (w.get("harvest_f0") or {}).get("ms", float("nan")), (w.get("harvest_f0") or {}).get("rtf", float("nan"))),
     "| 48 kHz: analysis / synthesis of %.0f s | %.2f / %.2f ms, RTF %.2e / %.2e |" % (w48["audio_seconds"], w48["analysis_ms"], w48["synthesis_ms"], w48["analysis_rtf"], w48["synthesis_rtf"]),
     "| MLPG, %d utterances, %d frames x 62 dims | %.2f ms, %.0f GB/s algorithmic (%.1f %% of HBM peak) |" % (ml["utterances"], ml["frames"], ml["ms"], ml["algorithmic_GBps"], 100 * ml["frac_of_hbm_peak"]),
     "| %s train step, %d utterances (%d valid frames) | %.1f ms -> %.0f k valid frames/s |" % (bl["model"], bl["utterances_per_gpu"], bl["valid_frames"], bl["ms_per_step"], bl["valid_frames_per_s"] / 1e3),
     "| %s train step, same batch | %.1f ms -> %.0f k valid frames/s |" % (bg["model"], bg["ms_per_step"], bg["valid_frames_per_s"] / 1e3),
     "| config 3 rooflines (fp32 MFMA 157.3 TF) | BiLSTM %.1f TFLOP/s = %.1f %%, BiGRU %.1f TFLOP/s = %.1f %% |" % (bl["roofline"]["achieved"], 100 * bl["roofline"]["frac"], bg["roofline"]["achieved"], 100 * bg["roofline"]["frac"]),
     "| config 4: duration model + MLPG inference, %d utterances | %.2f ms -> %.0f utterances/s (duration model alone %.3f ms, %.1f M phones/s) |" % (d["duration_mlpg"]["utterances"], d["duration_mlpg"]["ms"], d["duration_mlpg"]["utterances_per_s"], d["duration_mlpg"]["duration_model_ms"], d["duration_mlpg"]["phones_per_s"] / 1e6),
     "| `WorldFeatLabelGen.gen_data` end to end (wav files -> .npz + statistics, file I/O included), %d files = %.0f s | %.3f s -> RTF %.2e |" % (d["gen_data"]["utterances"], d["gen_data"]["audio_seconds"], d["gen_data"]["seconds"], d["gen_data"]["rtf"]),
     "| CPU baselines on this host (%s, %d logical CPUs) | C oracle analysis+synthesis, one process per core: RTF %.3f; torch-CPU BiLSTM stack: %.0f valid frames/s |" % (d["cpu_baseline"].get("cpu_model", "?"), d["cpu_baseline"].get("logical_cpus", 0), w.get("cpu_baseline_pool", {}).get("analysis_plus_synthesis_rtf", float("nan")), bl.get("cpu_baseline", {}).get("value", float("nan"))),
     "| epoch over an HBM-resident frame shard (%d utterances, %.2f M frames, %.1f GB) | %.1f ms -> %.1f M valid frames/s |" % (re_["utterances"], re_["frames"] / 1e6, re_["shard_GB"], re_["epoch_ms"], re_["valid_frames_per_s"] / 1e6),
     "\nFF-only run, top kernels (`%s_ff_kernel_stats.csv`):\n\n| kernel | calls | total ms | avg us | %% |\n|---|---|---|---|---|" % tag]
for r in ff[:9]:
    o.append("| `%s` | %s | %.2f | %.2f | %s |" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
o.append("\n`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline` (`%s_bench_kernel_stats.csv`; all sections of the bench in one process: FF loop, 16 kHz and 48 kHz WORLD passes, MLPG, BiLSTM / BiGRU steps, resident epoch):\n\n| kernel | calls | total ms | avg us | %% |\n|---|---|---|---|---|" % tag)
for r in full[:32]:
    o.append("| `%s` | %s | %.2f | %.2f | %s |" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(o) + "\n")
print("\n".join(o[:18]))
