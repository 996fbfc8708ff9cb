#!/bin/bash
# HBM traffic (PMC) of the synthesis sections after the per-kind pulse kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ah; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st; mkdir -p /tmp/st
for spec in "synthesis 16000" "synthesis 48000"; do
  set -- $spec; sec=$1; fs=$2
  for n in 3 5; do
    for c in FETCH_SIZE WRITE_SIZE; do
      d=/tmp/st/${sec}_${fs}_${n}_${c}
      rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/scripts/traffic_driver.py $sec $n $fs > $d.log 2>&1
    done
  done
done
python3 $R/scripts/traffic_summary.py /tmp/st $O/synthesis_section_traffic.json
python3 - <<PY
import json
d=json.load(open('$O/synthesis_section_traffic.json'))
for k,v in d['sections'].items(): print(k, {kk: (round(vv/1e9,3) if isinstance(vv,(int,float)) else {a: round(b/1e9,3) for a,b in vv.items()}) for kk,vv in v.items()})
PY
