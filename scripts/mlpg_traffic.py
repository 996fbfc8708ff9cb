"""HBM bytes per MLPG solve from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KB units) over
`MLPG_SIZES=256 python3 scripts/mlpg_curve.py stream fused multipass`: writes the per-path totals as
JSON.  gfx950 correction: FETCH_SIZE counts 64 B per 128-B request -> doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section).
usage: python3 scripts/mlpg_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import sys

GROUPS = {"stream": ("mlpg_prep_kernel", "mlpg_reduce_kernel", "mlpg_scan_kernel", "mlpg_solve_kernel"),
          "fused": ("mlpg_fused_kernel",),
          "multipass": ("mlpg_transfer_kernel", "mlpg_chunk_kernel")}


def totals(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    s, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for k in sum(GROUPS.values(), ()):
            if k in r["Kernel_Name"]:
                s[k] += float(r["Counter_Value"])
                n[k] += 1
    return s, n


fetch, nf = totals(sys.argv[1], "FETCH_SIZE")
write, nw = totals(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `MLPG_SIZES=256 python3 "
               "scripts/mlpg_curve.py stream fused multipass`; KB units, FETCH_SIZE doubled (gfx950: 64 B "
               "counted per 128-B request); 256 utterances, 310318 frames x 62 dims: algorithmic 620.6 MB "
               "per solve", "paths": {}}
for path, kernels in GROUPS.items():
    solves = None
    rd = wr = 0.0
    per_kernel = {}
    for k in kernels:
        assert nf[k] == nw[k], (k, nf[k], nw[k])
        if nf[k] == 0:
            continue
        launches_per_solve = {"mlpg_chunk_kernel": 4}.get(k, 1)
        solves = nf[k] // launches_per_solve
        r_, w_ = 2.0 * fetch[k] * 1024.0 / solves, write[k] * 1024.0 / solves
        per_kernel[k] = {"hbm_read_bytes_per_solve": r_, "hbm_write_bytes_per_solve": w_}
        rd += r_
        wr += w_
    if solves:
        out["paths"][path] = {"solves": solves, "hbm_read_bytes_per_solve": rd, "hbm_write_bytes_per_solve": wr,
                              "hbm_bytes_per_solve": rd + wr, "kernels": per_kernel}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
