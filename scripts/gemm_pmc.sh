# PMC passes over scripts/bench_gemm.py (run on the GPU box): bash scripts/gemm_pmc.sh <outdir>
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/scripts/bench_gemm.py > $out/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "gemm_f32" not in k: continue
        k = k + " grid=" + r["Grid_Size"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_BUSY_CU_CYCLES", "SQ_WAIT_INST_LDS", "SQ_WAVES", "SQ_INSTS_SALU"): cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        n = max(1, max(cnt.get((k, x), 0) for x in ("SQ_BUSY_CU_CYCLES", "SQ_WAIT_INST_LDS", "SQ_WAVES", "SQ_INSTS_SALU")))
        print("   %-32s %14.0f per launch" % (c, v / n))
PY
