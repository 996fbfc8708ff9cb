"""Per-kernel issue / LDS / wait fractions from SQ counters (rocprofv3 --pmc passes below a directory).

  valu_issue  = 4 x SQ_ACTIVE_INST_VALU / (kernel cycles x 1024 SIMDs)   share of the chip's VALU issue slots in use
  lds_busy    = SQ_LDS_IDX_ACTIVE / (kernel cycles x 256 CUs)            share of the LDS arrays' cycles in use
  waiting     = SQ_WAIT_ANY / SQ_WAVE_CYCLES                               share of a wave's life parked on a counter
  non_fp64    = 1 - (ADD + MUL + FMA + TRANS f64) / SQ_INSTS_VALU          integer, move, select, convert share
  conflicts   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
kernel cycles = SQ_BUSY_CYCLES / 32 (the counter is summed over the 32 shader engines; quad-cycle counters x 4).
usage: python3 scripts/pmc_fractions.py <dir> <out.json> [kernel substring ...]"""
import collections
import csv
import glob
import json
import sys

root, out_path, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if subs and not any(s in k for s in subs):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "SQ_BUSY_CYCLES" not in m or m["SQ_BUSY_CYCLES"] <= 0:
        continue
    cyc = m["SQ_BUSY_CYCLES"] / 32.0
    f64 = sum(m.get(c, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                      "SQ_INSTS_VALU_TRANS_F64"))
    row = {"launches_seen": len(d["SQ_BUSY_CYCLES"]), "kernel_cycles": round(cyc),
           "valu_insts": round(m.get("SQ_INSTS_VALU", 0.0)),
           "valu_issue": 4.0 * m.get("SQ_ACTIVE_INST_VALU", 0.0) / (cyc * 1024.0),
           "lds_busy": m.get("SQ_LDS_IDX_ACTIVE", 0.0) / (cyc * 256.0),
           "waiting": m.get("SQ_WAIT_ANY", 0.0) / max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0),
           "non_fp64": 1.0 - f64 / max(m.get("SQ_INSTS_VALU", 1.0), 1.0),
           "conflicts": m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)}
    out[k[-48:]] = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in row.items()}
json.dump(out, open(out_path, "w"), indent=1)
for k, r in out.items():
    print("%-48s valu_issue %.2f  lds_busy %.2f  waiting %.2f  non_fp64 %.2f  conflicts %.2f  (%d VALU insts)" % (
        k, r["valu_issue"], r["lds_busy"], r["waiting"], r["non_fp64"], r["conflicts"], r["valu_insts"]))
