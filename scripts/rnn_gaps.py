"""GPU busy time inside the BiLSTM steps of a rocprofv3 kernel trace of `bench.py --world-utts 0`: span from
the first to the last LSTM recurrence kernel, time with at least one kernel running, the largest idle gaps.
usage: python3 scripts/rnn_gaps.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:])
               for r in csv.DictReader(open(f))))
idx = [i for i, r in enumerate(rows) if "rnn_persist_fwd_kernel<4>" in r[2] or "rnn_persist_bwd_kernel<4>" in r[2]]
a, b = idx[0], idx[-1]
seg = rows[a:b + 1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
ev = sorted([(r[0], 1) for r in seg] + [(r[1], -1) for r in seg])
busy, depth, last, gaps = 0, 0, t0, []
for t, d in ev:
    if depth > 0:
        busy += t - last
    elif t - last > 20000:
        gaps.append((t - last, last))
    depth += d
    last = t
n_steps = len([r for r in seg if "rnn_persist_fwd_kernel<4>" in r[2]]) / 3.0
print("LSTM window: %.1f steps, span %.2f ms (%.2f per step), some kernel running %.2f ms, idle %.2f ms (%.2f per step)" % (
    n_steps, (t1 - t0) / 1e6, (t1 - t0) / 1e6 / n_steps, busy / 1e6, (t1 - t0 - busy) / 1e6, (t1 - t0 - busy) / 1e6 / n_steps))
gaps.sort(reverse=True)
print("gaps > 20 us: %d, sum %.2f ms; largest (us):" % (len(gaps), sum(g for g, _ in gaps) / 1e6), [round(g / 1e3) for g, _ in gaps[:12]])
# what runs right after the largest gaps
import collections
by = collections.defaultdict(lambda: [0, 0])
for g, at in gaps:
    if g > 5e6:
        continue                      # set-up between the sections, not part of a step
    nxt = next(r for r in seg if r[0] >= at + g - 1)
    prv = max((r for r in seg if r[1] <= at + 1), key=lambda r: r[1])
    k = prv[2][-26:] + " -> " + nxt[2][-26:]
    by[k][0] += g
    by[k][1] += 1
for k, (ns, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %7.2f ms in %3d gaps: %s" % (ns / 1e6, n, k))
