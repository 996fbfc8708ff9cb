"""Kernel sequence of ONE step out of a rocprofv3 --kernel-trace csv: name, duration, gap to the previous
kernel -- the step is cut at the n-th launch of a marker kernel.
usage: python3 scripts/step_sequence.py <dir> <marker substring> <n-th occurrence> [min_us]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
marker, nth = sys.argv[2], int(sys.argv[3])
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[nth], idx[nth + 1]
prev_end = int(rows[a - 1]["End_Timestamp"]) if a > 0 else int(rows[a]["Start_Timestamp"])
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    if d >= min_us:
        print("%9.1f us  +%7.1f  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, d, r["Kernel_Name"][:90]))
    prev_end = e
print("step span %.2f ms" % ((prev_end - t0) / 1e6))
