#!/bin/bash
# Kernel timeline of one MLPG call at the bench size (256 utterances): start offsets, durations and the gaps between the
# launches of a call (GPU box): bash scripts/mlpg_timeline.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mt && rocprofv3 --kernel-trace --output-format csv -d /tmp/mt -- python3 $R/scripts/traffic_driver.py mlpg 6 > /tmp/mt.log 2>&1
python3 - <<'PY' | tee $O/r5_mlpg_timeline.txt
import csv, glob
f = glob.glob('/tmp/mt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'mlpg' in r['Kernel_Name'] or 'copy' in r['Kernel_Name'].lower() or 'fill' in r['Kernel_Name'].lower()]
last = rows[-8:]
t0 = int(last[0]['Start_Timestamp']); prev = None
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:70]))
    prev = e
PY
