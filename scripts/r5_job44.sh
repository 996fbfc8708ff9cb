#!/bin/bash
# per-kernel times of a config-3 step (BiLSTM), trace kept so that GEMM launches can be told apart by duration
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5an; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/bl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bl -- python3 $R/scripts/prof_bilstm.py 4 LSTM > /tmp/bl.log 2>&1
tail -1 /tmp/bl.log | cut -c1-300
python3 $R/scripts/kstats.py /tmp/bl 2>/dev/null | head -24 | tee $O/bilstm_kstats.txt
python3 - <<'PY' | tee $O/bilstm_last_step_trace.txt
import csv, glob
f = glob.glob('/tmp/bl/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last step: from the last but one adam-like boundary; simply print the last 140 kernels
last = rows[-150:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%10.1f %9.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90]))
PY
