"""Prints the top rows of a rocprofv3 `*kernel_stats.csv` found below a directory.
usage: python3 scripts/kstats.py <dir> [n]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
print(f, "total kernel ms %.2f" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6))
for r in rows[:n]:
    print("%-64s %6s %9.2f ms %9.2f us %6s%%" % (r["Name"][:64], r["Calls"],
          float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
