# HBM traffic of the persistent recurrence kernels (run on the GPU box): bash scripts/rnn_pmc.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export ONCE=1
rm -rf /tmp/rpf /tmp/rpw
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rpf -- python3 $R/scripts/exp_rnn_persist_bwd.py LSTM > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rpw -- python3 $R/scripts/exp_rnn_persist_bwd.py LSTM > /dev/null 2>&1
python3 - <<PY | tee $O/$1_rnn_traffic.txt
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
print("one bidirectional 512-unit LSTM layer, 64 utterances, T = 1981 (KB units; FETCH_SIZE doubled: gfx950 counts 64 B per 128-B request)")
for k in f:
    print("%-42s launches %d  HBM read %.1f MB  write %.1f MB per launch" % (k, nf[k], 2 * f[k] / 1024 / nf[k], w[k] / 1024 / max(nw[k], 1)))
PY
