"""Kernels (and copies) after the last pause of more than 30 ms in a rocprofv3 --kernel-trace [--memory-copy-trace]
output directory: start, duration, gap to the previous record, name."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 30e6:
        cut = i
t0, prev, busy = rows[cut][0], rows[cut][0], 0
for s, e, name in rows[cut:]:
    print("%9.1f us %8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name[:90]))
    busy += e - s
    prev = max(prev, e)
print("records %d  span %.1f us  busy %.1f us" % (len(rows) - cut, (prev - t0) / 1e3, busy / 1e3))
