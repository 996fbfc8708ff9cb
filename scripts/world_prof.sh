# Per-kernel time of one WORLD analysis + synthesis pass at the bench size (run on the GPU box):
# bash scripts/world_prof.sh [fs] [utts]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --bilstm-utts 0 --no-cpu-baseline --world-utts ${2:-256} --world-fs ${1:-16000} > /tmp/wp.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/world_timeline.py /tmp/wp; python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/wp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last analysis pass = the kernels between the last two cheaptrick launches' surroundings: simply
# aggregate per kernel name and divide by the number of cheaptrick<false> launches of the 16 kHz batch
by = collections.defaultdict(list)
for r in rows:
    by[r['Kernel_Name'].split('(')[0][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sorted(((sum(v), k, len(v)) for k, v in by.items()), reverse=True)
for s, k, n in tot[:40]:
    print("%-72s n=%5d total %9.1f us  avg %9.1f us" % (k, n, s, s / n))
PY
