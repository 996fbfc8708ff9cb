"""HBM bytes per GEMM launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KB units) over
the FF bench: writes profiles/<tag>_gemm_traffic.json.  gfx950 correction: FETCH_SIZE counts 64 B
per 128-B request -> doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section).
usage: python3 scripts/gemm_traffic.py <fetch_dir> <write_dir> <out.json> "<command line>" """
import csv
import glob
import json
import sys


def total(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    s, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if ("gemm_ring" in r["Kernel_Name"] or "gemm_f32_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"])
            n += 1
    return s, n, f


fetch, n, ff = total(sys.argv[1], "FETCH_SIZE")
write, n2, wf = total(sys.argv[2], "WRITE_SIZE")
assert n == n2 and n > 0
out = {"kernel": "gemm_ring_kernel + gemm_ring_pair_kernel (the GEMM launches of the FF step)", "launches": n, "fetch_size_kb_sum": fetch,
       "write_size_kb_sum": write,
       "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 / n,
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `" + sys.argv[4] +
               "`; gfx950 correction: FETCH_SIZE counts 64 B per 128-B request -> doubled "
               "(MI355X_MICROARCH.md HBM section)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
