# PMC passes over the WORLD section of bench.py (run on the GPU box): bash scripts/world_pmc.sh <outdir under the repo>
# Counters of the FFT / solve kernels named in VERDICT r2 #5; the program goes directly after `--`.
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --bilstm-utts 0 --no-cpu-baseline --world-utts 64 > $out/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
want = ("syn_pulse_wave_kernel", "mcls_solve_dpp", "gemm_f64_kernel", "d4c_kernel", "cheaptrick_wave_kernel", "gemm_f64_lds")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(w in k for w in want): continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in sorted(agg.items()):
        line = k
        print(line); fo.write(line + "\n")
        for c, v in sorted(d.items()):
            line = "   %-30s n=%4d mean per launch %16.0f" % (c, len(v), sum(v) / len(v))
            print(line); fo.write(line + "\n")
PY
rm -rf $out/p*/*/*counter_collection.csv.bak
