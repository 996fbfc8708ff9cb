#!/bin/bash
# per-kernel statistics of the four WORLD sections at the head of the branch
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5ax; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for sec in analysis synthesis; do
  for fs in 16000 48000; do
    rm -rf /tmp/ak; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ak -- python3 $R/scripts/traffic_driver.py $sec 4 $fs > /tmp/ak.log 2>&1
    python3 $R/scripts/kstats.py /tmp/ak 14 2>/dev/null > $O/${sec}_${fs}_kstats.txt
    head -6 $O/${sec}_${fs}_kstats.txt
  done
done
