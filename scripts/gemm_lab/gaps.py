"""Per-kernel duration and the gap to the previous kernel from a rocprofv3 kernel trace.
usage: python3 scripts/gemm_lab/gaps.py <dir>"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
stat = collections.OrderedDict()
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:70]
    d = stat.setdefault(name, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
    if prev_end is not None:
        d[2] += (s - prev_end) / 1e3
    prev_end = e
for k, (n, dur, gap) in stat.items():
    print("%-70s n=%4d  dur %8.1f us  gap before %6.1f us" % (k, n, dur / n, gap / n))
