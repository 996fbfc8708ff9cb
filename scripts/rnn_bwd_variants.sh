# Diagnostic for the backward recurrence's HBM write traffic (run on the GPU box):
#   bash scripts/rnn_bwd_variants.sh <tag>
# Builds lstm.hip with the exchange tiles at a pitch of 88 (product), 64 and 32 granules (the smaller
# pitches make neighbouring tiles overlap: WRONG numerics, same instruction stream -- they only show how
# WRITE_SIZE and the step time depend on the exchange footprint per XCD: 2.75 / 2.0 / 1.0 MB).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=$R/idiaptts_amd/_lib
cp $L/libidiaptts_amd.so /tmp/lib_orig.so
OBJS=$(ls $L/*.o | grep -v lstm.o)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for PT in 88 64 32; do
  hipcc $FLAGS -DPERSIST_BWD_PT=$PT -c $R/idiaptts_amd/csrc/lstm.hip -o /tmp/lstm_$PT.o || exit 1
  hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $OBJS /tmp/lstm_$PT.o || exit 1
  echo "== PT $PT" | tee -a $O/$1_rnn_bwd_variants.txt
  python3 $R/scripts/exp_rnn_persist_bwd.py LSTM 2>&1 | tail -1 | tee -a $O/$1_rnn_bwd_variants.txt
  rm -rf /tmp/rpf /tmp/rpw
  ONCE=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rpf -- python3 $R/scripts/exp_rnn_persist_bwd.py LSTM > /dev/null 2>&1
  ONCE=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rpw -- python3 $R/scripts/exp_rnn_persist_bwd.py LSTM > /dev/null 2>&1
  python3 - <<PY | tee -a $O/$1_rnn_bwd_variants.txt
import csv, glob, collections
def tot(d, c):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    s, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "persist" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][-40:]
            s[k] += float(r["Counter_Value"]); n[k] += 1
    return s, n
f, nf = tot("/tmp/rpf", "FETCH_SIZE"); w, nw = tot("/tmp/rpw", "WRITE_SIZE")
for k in f:
    print("%-42s launches %d  HBM read %.1f MB  write %.1f MB per launch" % (k, nf[k], 2 * f[k] / 1024 / nf[k], w[k] / 1024 / max(nw[k], 1)))
PY
done
cp /tmp/lib_orig.so $L/libidiaptts_amd.so
