"""Mean of each PMC counter per kernel name from rocprofv3 `*counter_collection.csv` files below a
directory.  usage: python3 scripts/pmc_summary.py <dir> [kernel substring]"""
import collections
import csv
import glob
import sys

sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("    %-28s n=%4d mean %14.3f" % (c, len(v), sum(v) / len(v)))
