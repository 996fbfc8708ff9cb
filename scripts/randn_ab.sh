# kernel statistics of the world section, the noise kernels' rows (round 6: randn_u32_kernel's stores through an LDS tile,
# 930 -> 639 us; the same change to syn_randn_kernel left its 140 us where they were -- integer-ALU bound -- and was dropped)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_w -- python3 $GRAFT_REPO_ROOT/scripts/run_section.py world 256 16000 25.0 False False > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/scripts/kstats.py /tmp/ks_w 40 | grep -i "randn\|total"
