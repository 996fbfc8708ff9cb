"""Per-section HBM bytes from the PMC passes of scripts/section_traffic.sh.
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced read and is doubled here (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as it is.
usage: python3 scripts/traffic_summary.py <dir> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

root, out_path = sys.argv[1], sys.argv[2]


def total(d, counter):
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    if not files:
        return None, {}
    tot, per = 0.0, collections.defaultdict(float)
    for r in csv.DictReader(open(files[-1])):
        if r["Counter_Name"] == counter:
            v = float(r["Counter_Value"]) * 1024.0
            tot += v
            per[r["Kernel_Name"].split("(")[0][-60:]] += v
    return tot, per


out = {"note": "HBM bytes per pass (WORLD sections, MLPG) or per training step (bilstm, bigru): rocprofv3 --pmc "
               "FETCH_SIZE and --pmc WRITE_SIZE in separate runs at 3 and at 5 passes of scripts/traffic_driver.py, "
               "(counters at 5 - counters at 3) / 2; FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request)",
       "sections": {}}
names = sorted({os.path.basename(p)[:-len("_FETCH_SIZE")].rsplit("_", 1)[0] for p in glob.glob(root + "/*_FETCH_SIZE")})
for name in names:
    row = {}
    kern = collections.defaultdict(float)
    for counter, scale, key in (("FETCH_SIZE", 2.0, "read_bytes"), ("WRITE_SIZE", 1.0, "write_bytes")):
        lo, plo = total("%s/%s_3_%s" % (root, name, counter), counter)
        hi, phi = total("%s/%s_5_%s" % (root, name, counter), counter)
        if lo is None or hi is None:
            continue
        row[key] = scale * (hi - lo) / 2.0
        for k in phi:
            kern[k] += scale * (phi[k] - plo.get(k, 0.0)) / 2.0
    if row:
        row["hbm_bytes"] = row.get("read_bytes", 0.0) + row.get("write_bytes", 0.0)
        row["largest_kernels"] = {k: round(v) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])[:6]}
        sec, fs = name.rsplit("_", 1)
        out["sections"][sec if fs == "0" else "%s_%s" % (sec, fs)] = row
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: {"GB": round(v["hbm_bytes"] / 1e9, 3)} for k, v in out["sections"].items()}))
