"""Coarse timeline of a rocprofv3 kernel trace: consecutive launches of the same kernel merged.
usage: python3 scripts/timeline.py <dir> [t_from_ms] [t_to_ms]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
cur = None
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if s < lo or s > hi:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("itts::", "").replace("void ", "")[:40]
    key = (name, r["Stream_Id"])
    if cur and cur[0] == key:
        cur[2] = e
        cur[3] += 1
        cur[4] += e - s
    else:
        if cur:
            print("%9.3f - %9.3f ms  stream %-3s %-42s x%-4d busy %8.3f ms" % (cur[1], cur[2], cur[0][1], cur[0][0], cur[3], cur[4]))
        cur = [key, s, e, 1, e - s]
if cur:
    print("%9.3f - %9.3f ms  stream %-3s %-42s x%-4d busy %8.3f ms" % (cur[1], cur[2], cur[0][1], cur[0][0], cur[3], cur[4]))
