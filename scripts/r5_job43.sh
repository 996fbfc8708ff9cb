#!/bin/bash
# lab: d4c_kernel<false, 11> at four workgroups per CU (128 registers: 64 spilled; the LDS claim cut to 40 KB for the
# timing only -- results invalid) against the kernel as it is
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5am; mkdir -p $O
L=idiaptts_amd/_lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
cd /tmp; export TMPDIR=/tmp
for def in "-DD4C_OCC=4" ""; do
  ( cd $R && /opt/rocm/bin/hipcc $FLAGS $def -c idiaptts_amd/csrc/world_f0ap.hip -o $L/world_f0ap.o 2>/dev/null && /opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $L/*.o ) || exit 3
  echo "== ${def:-as it is}" | tee -a $O/d4c_occ.txt
  if [ -n "$def" ]; then export ITTS_D4C_LAB_LDS=40960; else unset ITTS_D4C_LAB_LDS; fi
  rm -rf /tmp/ak; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ak -- python3 $R/scripts/traffic_driver.py analysis 4 16000 > /tmp/ak.log 2>&1
  python3 $R/scripts/kstats.py /tmp/ak 2>/dev/null | head -4 | tee -a $O/d4c_occ.txt
done
