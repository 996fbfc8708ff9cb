# Round-5 evidence (run on the GPU box): bash scripts/prof_round5.sh <tag>   -> gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
FF="--world-utts 0 --bilstm-utts 0 --no-cpu-baseline --trainer-utts 0"
# 1. the driver's protocol, three times (the line the round is judged on), and the long form
for i in 1 2 3; do python3 $R/bench.py --steps 20 --warmup 5 $FF 2>> $O/ff_stderr.txt | tail -1 > $O/ff_bench_line_20_5_run$i.json; done
python3 $R/bench.py $FF 2>> $O/ff_stderr.txt | tail -1 > $O/ff_bench_line_200_20.json
# 2. per-kernel statistics of the FF step and of the whole default bench
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ff_raw -- python3 $R/bench.py $FF > /dev/null 2>&1
cp $(ls $O/ff_raw/*/*kernel_stats.csv | head -1) $O/ff_kernel_stats.csv; rm -rf $O/ff_raw
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full_raw -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --trainer-utts 0 > /dev/null 2>&1
cp $(ls $O/full_raw/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/full_raw
# 3. HBM traffic: the GEMM launches of the FF step, then every secondary section
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
python3 $R/scripts/gemm_traffic.py $O/pmc_f $O/pmc_w $O/gemm_traffic.json "bench.py --steps 20 --warmup 3 $FF" > $O/gemm_traffic.txt 2>&1
rm -rf $O/pmc_f $O/pmc_w
bash $R/scripts/section_traffic.sh $1 > $O/section_traffic.txt 2>&1
mv $R/gpurun_out/$1_section_traffic.json $O/section_traffic.json
# 4. matrix-unit counters: the six launches of the FF step, then one BiLSTM / BiGRU training step
#    SQ_INSTS_MFMA: MFMA instructions issued; SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES: share of the CUs' busy
#    cycles in which their matrix units were busy (both summed over the chip by the profiler)
rm -rf /tmp/mf5
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/mf5/ff -- python3 $R/bench.py --steps 20 --warmup 3 $FF > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/mf5/lstm -- python3 $R/scripts/traffic_driver.py bilstm 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/mf5/gru -- python3 $R/scripts/traffic_driver.py bigru 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/mf5/ana -- python3 $R/scripts/traffic_driver.py analysis 2 16000 64 > /dev/null 2>&1
python3 - /tmp/mf5 > $O/mfma_busy.txt <<'PY'
import collections, csv, glob, sys
for sec in ("ff", "lstm", "gru", "ana"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[1] + "/" + sec + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not any(w in k for w in ("gemm_ring", "rnn_persist", "mcls_fused3", "mcls_init_fused", "gemm_f64")):
                continue
            agg[k[:86]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== " + sec)
    for k, d in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(m.get("SQ_BUSY_CU_CYCLES", 1.0), 1.0)
        print("%-86s launches %4d  SQ_INSTS_MFMA %12.0f  SQ_INSTS_VALU %12.0f  MFMA_BUSY / BUSY_CU %.3f" % (
            k, len(d.get("SQ_INSTS_MFMA", [])), m.get("SQ_INSTS_MFMA", 0.0), m.get("SQ_INSTS_VALU", 0.0), busy))
PY
# 5. issue / LDS / wait fractions of the WORLD kernels (64 utterances: the counters serialise the kernels)
rm -rf /tmp/wp5; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/wp5/a$i -- python3 $R/scripts/traffic_driver.py analysis 2 16000 64 > /dev/null 2>&1
  rocprofv3 --pmc $set --output-format csv -d /tmp/wp5/s$i -- python3 $R/scripts/traffic_driver.py synthesis 2 16000 64 > /dev/null 2>&1
done
python3 $R/scripts/pmc_fractions.py /tmp/wp5 $O/world_pmc_fractions.json syn_pulse d4c_kernel cheaptrick mcls_ mgc2sp gemm_f64 stonemask dio_ > $O/world_pmc_fractions.txt 2>&1
# 6. per-kernel time of one analysis and one synthesis at the bench sizes
SERIAL=1 bash $R/scripts/analysis_prof.sh $1 256 16000 > $O/analysis_16k.txt 2>&1
SERIAL=1 bash $R/scripts/analysis_prof.sh $1 64 48000 > $O/analysis_48k.txt 2>&1
rm -rf /tmp/sp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python3 $R/scripts/traffic_driver.py synthesis 4 16000 256 > /dev/null 2>&1
python3 $R/scripts/kstats.py /tmp/sp 20 > $O/synthesis_16k_kstats.txt 2>&1
# 7. recurrences: traffic of one layer
bash $R/scripts/rnn_pmc.sh $1 > $O/rnn_pmc.txt 2>&1
mv $R/gpurun_out/$1_rnn_traffic.txt $O/rnn_traffic.txt
ls -la $O
