# One steady-state step of the default module path (AcousticModelTrainer.train, device batch cache) as a kernel timeline:
# which launches a step is made of, in order, with full names.   gpurun -- 'bash scripts/module_path_step_trace.sh <tag>'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-mp_trace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ITTS_TRAINER_EPOCH_ONLY=${ITTS_TRAINER_EPOCH_ONLY:-module_path}
rm -rf /tmp/mpt && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/mpt -- python3 $R/scripts/run_trainer_epoch.py > $O/epoch.txt 2>&1
python3 - <<'PY' > $O/step_timeline.txt
import csv, glob
rows = []
for f in glob.glob("/tmp/mpt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob("/tmp/mpt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
# the last but 3rd adam launch .. the next one: one training step of the last epoch
adam = [i for i, r in enumerate(rows) if "adam" in r[2]]
a, b = adam[-5], adam[-4]
t0 = rows[a][1]; prev = t0; busy = 0
for s, e, name in rows[a + 1:b + 1]:
    print("%9.1f us %8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name[:200]))
    busy += e - s; prev = max(prev, e)
print("records %d  span %.1f us  busy %.1f us" % (b - a, (prev - t0) / 1e3, busy / 1e3))
PY
tail -5 $O/epoch.txt
