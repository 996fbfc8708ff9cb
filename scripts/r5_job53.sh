#!/bin/bash
# kernel timeline of one 16 kHz analysis pass (both streams): where are the gaps and the small launches?
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5az; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/at; rocprofv3 --kernel-trace --output-format csv -d /tmp/at -- python3 $R/scripts/traffic_driver.py analysis 3 16000 > /tmp/at.log 2>&1
python3 - <<'PY' > $O/analysis_timeline.txt
import csv, glob
f = glob.glob('/tmp/at/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last pass: from the last dio_lowcut_kernel on
idx = max(i for i, r in enumerate(rows) if 'dio_lowcut' in r['Kernel_Name'])
idx = max(0, idx - 4)
t0 = int(rows[idx]['Start_Timestamp'])
prev_end = {}
for r in rows[idx:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r.get('Queue_Id', '?')
    gap = (s - prev_end.get(q, s)) / 1e3
    prev_end[q] = e
    print("%9.1f us  %8.1f us  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r['Kernel_Name'][:70]))
PY
wc -l $O/analysis_timeline.txt; awk '{ if ($6+0 > 15.0) print }' $O/analysis_timeline.txt | head -40
cd $R; timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_properties.py tests/test_gpu_dropin.py tests/test_gpu_mgc.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --steps 5 --warmup 2 --ramp-steps 0 --no-cpu-baseline --bilstm-utts 0 --trainer-utts 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench: 16k analysis %.3f synthesis %.3f ms  48k analysis %.3f synthesis %.3f ms gen_data %.3g' % (j['world']['analysis_ms'], j['world']['synthesis_ms'], j['world_48k']['analysis_ms'], j['world_48k']['synthesis_ms'], j['gen_data']['rtf']))"
