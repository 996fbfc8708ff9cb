"""Per-launch duration of the recurrence step kernels by grid size, from a rocprofv3 kernel trace.
usage: python3 scripts/step_hist.py <dir> [name-substring]"""
import collections
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
sub = sys.argv[2] if len(sys.argv) > 2 else "_step_"
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if sub in n:
        wg = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
        d[(n.split('(')[0].split('::')[-1], wg, int(r['Workgroup_Size_X']))].append(
            int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k in sorted(d):
    v = sorted(d[k])
    print("%-28s WGs/dir %5d x %4d thr  n=%6d  mean %6.2f us  median %6.2f  min %6.2f" % (
        k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3))
