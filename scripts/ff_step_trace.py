"""Per-launch durations of one FF train step from a rocprofv3 kernel trace CSV
(rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --world-utts 0 --bilstm-utts 0
--no-cpu-baseline --steps 40 --warmup 5): the launches of the steady-state steps in issue order,
median duration per position, with the GEMM shapes of the 425-512-512-187 model beside them.
usage: python3 scripts/ff_step_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts at every first-layer forward GEMM: find the period by the adam kernel
adam = [i for i, n in enumerate(names) if "adam" in n]
per = int(np.median(np.diff(adam)))
steps = [(a - per + 1, a + 1) for a in adam if a - per + 1 >= 0]
steps = [s for s in steps if names[s[1] - 1].startswith(names[adam[-1]][:10])][5:-2]
dur = defaultdict(list)
gap = defaultdict(list)
for a, b in steps:
    for k, i in enumerate(range(a, b)):
        dur[k].append(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]))
        if i > a:
            gap[k].append(int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]))
a, b = steps[len(steps) // 2]
tot = 0.0
for k, i in enumerate(range(a, b)):
    d = np.median(dur[k]) / 1e3
    g = np.median(gap[k]) / 1e3 if gap[k] else 0.0
    tot += d + g
    print("%2d %-70s %8.1f us  gap %5.1f" % (k, names[i][:70], d, g))
print("step total %.1f us over %d steps" % (tot, len(steps)))
