set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; mkdir -p $O/ff $O/full $O/pmc_f $O/pmc_w
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -1 $O/bench_stdout.txt > $O/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ff_raw -- python3 $R/bench.py --world-utts 0 --bilstm-utts 0 --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/ff_raw/*/*kernel_stats.csv | head -1) $O/ff/ff_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full_raw -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/full_raw/*/*kernel_stats.csv | head -1) $O/full/full_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 20 --warmup 3 --world-utts 0 --bilstm-utts 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 20 --warmup 3 --world-utts 0 --bilstm-utts 0 --no-cpu-baseline > /dev/null 2>&1
rm -rf $O/ff_raw/*/*kernel_trace.csv $O/full_raw/*/*kernel_trace.csv
ls -la $O $O/pmc_f/* | head -30
