# Round-4 evidence (run on the GPU box): bash scripts/prof_round4.sh <tag>   -> gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
FF="--world-utts 0 --bilstm-utts 0 --no-cpu-baseline"
# 1. the driver's protocol, three times (the line the round is judged on), and the long form
for i in 1 2 3; do python3 $R/bench.py --steps 20 --warmup 5 $FF 2>> $O/ff_stderr.txt | tail -1 > $O/ff_bench_line_20_5_run$i.json; done
python3 $R/bench.py $FF 2>> $O/ff_stderr.txt | tail -1 > $O/ff_bench_line_200_20.json
# 2. per-kernel statistics of the FF step and of the whole default bench
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ff_raw -- python3 $R/bench.py $FF > /dev/null 2>&1
cp $(ls $O/ff_raw/*/*kernel_stats.csv | head -1) $O/ff_kernel_stats.csv; rm -rf $O/ff_raw
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full_raw -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $(ls $O/full_raw/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/full_raw
# 3. HBM traffic: the GEMM launches of the FF step, then every secondary section
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
python3 $R/scripts/gemm_traffic.py $O/pmc_f $O/pmc_w $O/gemm_traffic.json "bench.py --steps 20 --warmup 3 $FF" > $O/gemm_traffic.txt 2>&1
rm -rf $O/pmc_f $O/pmc_w
bash $R/scripts/section_traffic.sh $1 > $O/section_traffic.txt 2>&1
mv $R/gpurun_out/$1_section_traffic.json $O/section_traffic.json
# 4. issue / LDS / wait fractions of the WORLD kernels (64 utterances: the counters serialise the kernels)
rm -rf /tmp/wp4
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/wp4/a$i -- python3 $R/scripts/traffic_driver.py analysis 2 16000 64 > /dev/null 2>&1
  rocprofv3 --pmc $set --output-format csv -d /tmp/wp4/s$i -- python3 $R/scripts/traffic_driver.py synthesis 2 16000 64 > /dev/null 2>&1
done
python3 $R/scripts/pmc_fractions.py /tmp/wp4 $O/world_pmc_fractions.json syn_pulse d4c_kernel cheaptrick mcls_ mgc2sp gemm_f64 stonemask dio_ > $O/world_pmc_fractions.txt 2>&1
# 5. per-kernel time of one analysis and one synthesis at the bench sizes
bash $R/scripts/analysis_prof.sh $1 256 16000 > $O/analysis_16k.txt 2>&1
bash $R/scripts/analysis_prof.sh $1 64 48000 > $O/analysis_48k.txt 2>&1
rm -rf /tmp/sp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python3 $R/scripts/traffic_driver.py synthesis 4 16000 256 > /dev/null 2>&1
python3 $R/scripts/kstats.py /tmp/sp 20 > $O/synthesis_16k_kstats.txt 2>&1
# 6. recurrences: traffic of one layer
bash $R/scripts/rnn_pmc.sh $1 > $O/rnn_pmc.txt 2>&1
mv $R/gpurun_out/$1_rnn_traffic.txt $O/rnn_traffic.txt
ls -la $O
