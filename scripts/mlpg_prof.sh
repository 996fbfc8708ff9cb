cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -- python3 $GRAFT_REPO_ROOT/scripts/mlpg_curve.py ${MODES:-stream fused} > /tmp/mp.log 2>&1
grep utts /tmp/mp.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/mp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'mlpg' in n:
        by[n.split('(')[0][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in by.items():
    v3 = v[-6:]
    print("%-62s n=%3d last6 mean %9.1f us  (all: min %.1f max %.1f)" % (n, len(v), sum(v3) / len(v3), min(v), max(v)))
PY
