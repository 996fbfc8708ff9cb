"""Timeline of the LAST WORLD analysis pass found in a rocprofv3 kernel trace (csv): span, time with
at least one kernel running, per-kernel totals inside the window.
usage: python3 scripts/world_timeline.py <dir with *kernel_trace.csv> [true|false [pass index]]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]) for r in csv.DictReader(open(f))]
rows.sort()
# an analysis pass starts with dio_lowcut_kernel and ends with the last mcls_* kernel before the next
# dio_lowcut / syn_* kernel
starts = [i for i, r in enumerate(rows) if 'dio_lowcut' in r[2]]
passes = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    seg = rows[a:b]
    end = max((i for i, r in enumerate(seg) if 'mcls' in r[2] or 'd4c' in r[2]), default=None)
    if end is not None:
        passes.append(seg[:end + 1])
for q in passes:
    print("pass: %3d launches, span %7.2f ms, first %s ... %s" % (
        len(q), (max(r[1] for r in q) - q[0][0]) / 1e6, q[0][2][:24],
        "d4c<true>" if any("d4c_kernel<true>" in r[2] for r in q) else "d4c<false>"))
# the pass looked at: the last one of the requested kind (argv[2]: "true" / "false" = d4c_kernel<...>)
kind = "d4c_kernel<%s>" % (sys.argv[2] if len(sys.argv) > 2 else "true")
p = [q for q in passes if any(kind in r[2] for r in q)][-1]
if len(sys.argv) > 3:
    p = passes[int(sys.argv[3])]
t0, t1 = p[0][0], max(r[1] for r in p)
ev = sorted([(r[0], 1) for r in p] + [(r[1], -1) for r in p])
busy, depth, last = 0, 0, t0
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print("pass span %.2f ms, some kernel running %.2f ms, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, len(p)))
by = collections.defaultdict(lambda: [0, 0])
for s, e, k in p:
    by[k][0] += e - s
    by[k][1] += 1
for k, (ns, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:22]:
    print("  %-60s n=%4d %8.2f ms" % (k, n, ns / 1e6))
# gaps on the timeline (no kernel running) longer than 50 us
depth, last, gaps = 0, t0, []
for t, d in ev:
    if depth == 0 and t - last > 50000:
        gaps.append(((last - t0) / 1e6, (t - last) / 1e3))
    depth += d
    last = t
print("idle gaps > 50 us (at ms, length us):", [(round(a, 2), round(b)) for a, b in gaps][:30])
