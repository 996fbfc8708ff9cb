# Round-6 evidence (run on the GPU box): bash scripts/prof_round6.sh <tag>   -> gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
FF="--world-utts 0 --bilstm-utts 0 --no-cpu-baseline --trainer-utts 0"
# 1. the driver's protocol, three times (the line the round is judged on)
for i in 1 2 3; do python3 $R/bench.py --steps 20 --warmup 5 $FF 2>> $O/ff_stderr.txt | tail -1 > $O/ff_bench_line_20_5_run$i.json; done
# 2. per-kernel statistics of the FF step (the same command under the profiler)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ff_raw -- python3 $R/bench.py $FF > /dev/null 2>&1
cp $(ls $O/ff_raw/*/*kernel_stats.csv | head -1) $O/ff_kernel_stats.csv; rm -rf $O/ff_raw
# 3. HBM traffic of the GEMM launches of the FF step (separate passes, FETCH_SIZE doubled: the guide's gfx950 correction)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
python3 $R/scripts/gemm_traffic.py $O/pmc_f $O/pmc_w $O/gemm_traffic.json "bench.py --steps 20 --warmup 3 $FF" > $O/gemm_traffic.txt 2>&1
rm -rf $O/pmc_f $O/pmc_w
# 4. the module path from the device batch cache: kernels of three epochs of AcousticModelTrainer.train, and the
#    traffic of the batch kernels against their algorithmic bytes
export ITTS_TRAINER_EPOCH_ONLY=module_path
rm -rf /tmp/mp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -- python3 $R/scripts/run_trainer_epoch.py > $O/module_path_epoch.txt 2>&1
python3 $R/scripts/kstats.py /tmp/mp 30 > $O/module_path_kstats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/mp_$c && rocprofv3 --pmc $c --output-format csv -d /tmp/mp_$c -- python3 $R/scripts/run_trainer_epoch.py > /dev/null 2>&1
done
python3 - > $O/batch_rows_traffic.txt <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("/tmp/mp_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "batch_" in k and r["Counter_Name"] == c:
                agg[k][c].append(float(r["Counter_Value"]))
print("HBM traffic per launch of the batch kernels during three epochs of the cached module path (1 024 utterances, batches of 32;")
print("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KB; FETCH_SIZE doubled as the guide prescribes for gfx950)")
for k, d in sorted(agg.items()):
    f, w = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
    print("%-60s launches %4d  fetch %9.1f MB  write %9.1f MB per launch" % (k[:60], len(f), 2 * sum(f) / len(f) / 1024, sum(w) / max(len(w), 1) / 1024))
PY
unset ITTS_TRAINER_EPOCH_ONLY
# 5. one utterance through the one-utterance API: kernel timelines
for what in analysis synthesis mlpg; do
  rm -rf /tmp/tr && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python3 $R/scripts/single_call_driver.py $what > /tmp/tr.log 2>&1
  python3 $R/scripts/trace_tail.py /tmp/tr > $O/single_call_${what}_timeline.txt 2>&1
done
# 6. the whole default bench (what the driver runs), its line and its kernel statistics
python3 $R/bench.py --steps 20 --warmup 5 2>> $O/bench_stderr.txt | tail -1 > $O/bench_line_20_5.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full_raw -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/full_raw/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/full_raw
ls -la $O
