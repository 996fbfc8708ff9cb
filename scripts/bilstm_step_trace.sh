# Kernel timeline of the last config-3 training step (scripts/prof_bilstm.py under rocprofv3 --kernel-trace): every launch of the
# step in order, with the gaps.   gpurun -- 'bash scripts/bilstm_step_trace.sh <tag> [LSTM|GRU]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-bl_trace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/blt && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/blt -- python3 $R/scripts/prof_bilstm.py 3 ${2:-LSTM} > $O/section.txt 2>&1
python3 - <<'PY' > $O/step_timeline.txt
import csv, glob
rows = []
for f in glob.glob("/tmp/blt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob("/tmp/blt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam" in r[2]]
a, b = adam[-2], adam[-1]
t0 = rows[a][1]; prev = t0; busy = 0; small = 0; nsmall = 0
for s, e, name in rows[a + 1:b + 1]:
    print("%9.1f us %8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name[:150]))
    busy += e - s; prev = max(prev, e)
    if e - s < 100e3: small += e - s; nsmall += 1
print("records %d  span %.1f us  busy %.1f us  launches under 100 us: %d, %.1f us" % (b - a, (prev - t0) / 1e3, busy / 1e3, nsmall, small / 1e3))
PY
tail -3 $O/step_timeline.txt
