#!/bin/bash
# Kernel timeline of one synthesis pass at the bench size (GPU box): bash scripts/syn_timeline.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/syt && rocprofv3 --kernel-trace --output-format csv -d /tmp/syt -- python3 $R/scripts/syn_timeline.py 4 > /tmp/syt.log 2>&1
tail -2 /tmp/syt.log
python3 - <<'PY' | tee $O/r5_syn_timeline.txt
import csv, glob
f = glob.glob('/tmp/syt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last pass: from the last syn_inc_kernel (or the gemm in front of it) on
idx = max(i for i, r in enumerate(rows) if 'syn_inc_kernel' in r['Kernel_Name'])
start = max(0, idx - 6)
last = rows[start:]
t0 = min(int(r['Start_Timestamp']) for r in last if int(r['Start_Timestamp']) >= int(rows[idx]['Start_Timestamp']) - 200000)
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < t0: continue
    print("%9.1f us .. %9.1f us  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:64]))
PY
