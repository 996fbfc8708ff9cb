# Round-3 evidence (run on the GPU box): bash scripts/prof_round3.sh <tag>   -> gpurun_out/<tag>/
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
FF="--world-utts 0 --bilstm-utts 0 --no-cpu-baseline"
python3 $R/bench.py $FF 2> $O/ff_stderr.txt | tail -1 > $O/ff_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ff_raw -- python3 $R/bench.py $FF > /dev/null 2>&1
cp $(ls $O/ff_raw/*/*kernel_stats.csv | head -1) $O/ff_kernel_stats.csv
python3 $R/scripts/gemm_lab/gaps.py $O/ff_raw > $O/ff_gaps.txt 2>&1
rm -rf $O/ff_raw
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
python3 $R/scripts/gemm_traffic.py $O/pmc_f $O/pmc_w $O/gemm_traffic.json "bench.py --steps 20 --warmup 3 $FF" > $O/gemm_traffic.txt 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
python3 $R/scripts/pmc_summary.py $O/pmc_m gemm_ring > $O/gemm_mfma_busy.txt 2>&1
find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O
# the whole default bench: line + per-kernel statistics of every section
python3 $R/bench.py 2> $O/full_stderr.txt | tail -1 > $O/full_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full_raw -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/full_raw/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
rm -rf $O/full_raw
# MLPG: HBM traffic of the solves
bash $R/scripts/mlpg_pmc.sh $1 > $O/mlpg_pmc.txt 2>&1
mv $R/gpurun_out/$1_mlpg_traffic.json $O/mlpg_traffic.json
ls -la $O
