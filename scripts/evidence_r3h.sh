R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > $O/r3h_pytest_tail.txt
timeout 900 python bench.py 2> $O/r3h_bench_stderr.txt | tail -1 > $O/r3h_bench_line.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/full_raw -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>&1
cp $(ls /tmp/full_raw/*/*kernel_stats.csv | head -1) $O/r3h_bench_kernel_stats.csv
cat $O/r3h_pytest_tail.txt; head -c 600 $O/r3h_bench_line.json
