#!/bin/bash
# HBM traffic (PMC) of the WORLD sections after the 48 kHz / pulse-kernel changes: as scripts/section_traffic.sh, WORLD rows only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st; mkdir -p /tmp/st
for spec in "analysis 16000" "synthesis 16000" "analysis 48000" "synthesis 48000"; do
  set -- $spec; sec=$1; fs=$2
  for n in 3 5; do
    for c in FETCH_SIZE WRITE_SIZE; do
      d=/tmp/st/${sec}_${fs}_${n}_${c}
      rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/scripts/traffic_driver.py $sec $n $fs > $d.log 2>&1
    done
  done
done
python3 $R/scripts/traffic_summary.py /tmp/st $O/r5t_world_section_traffic.json
