#!/bin/bash
# MLPG one-pass kernel after the head rework (hoisted constants, table rows prefetched on the way back, one bounds
# load, input rows non-temporal): parity tests, times, HBM traffic (PMC), kernel statistics
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5y; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mlpg.py tests/test_gpu_fuzz.py tests/test_gpu_dropin.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest.txt
cat $O/pytest.txt
for a in "100 256 f64" "30 1024 f64" "30 4096 f64" "100 256 f32" "30 4096 f32"; do
  timeout 300 python3 scripts/mlpg_time.py $a 2>&1 | tail -1 | tee -a $O/time.txt
done
for v in "1,1,1" "1,0.1,0.05" "1,0.03,0.03" "1,0.01,0.01" "1,0.003,0.003"; do
  echo -n "var $v  " | tee -a $O/time.txt
  MLPG_TIME_VAR=$v timeout 300 python3 scripts/mlpg_time.py 30 4096 f64 2>&1 | tail -1 | tee -a $O/time.txt
done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st; mkdir -p /tmp/st
for n in 3 5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/st/mlpg_0_${n}_${c}
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/scripts/traffic_driver.py mlpg $n 0 > $d.log 2>&1
  done
done
python3 $R/scripts/traffic_summary.py /tmp/st $O/mlpg_section_traffic.json
rm -rf /tmp/mk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mk -- python3 $R/scripts/traffic_driver.py mlpg 20 0 > /tmp/mk.log 2>&1
python3 $R/scripts/kstats.py /tmp/mk 2>/dev/null | head -12 > $O/mlpg_kstats.txt; cat $O/mlpg_kstats.txt
