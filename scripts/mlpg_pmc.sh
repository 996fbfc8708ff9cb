# HBM traffic of the MLPG solves (run on the GPU box): bash scripts/mlpg_pmc.sh <tag> -> gpurun_out/<tag>_mlpg_traffic.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export MLPG_SIZES=256
rm -rf /tmp/mpf /tmp/mpw
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/mpf -- python3 $R/scripts/mlpg_curve.py stream fused multipass > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/mpw -- python3 $R/scripts/mlpg_curve.py stream fused multipass > /dev/null 2>&1
python3 $R/scripts/mlpg_traffic.py /tmp/mpf /tmp/mpw $O/$1_mlpg_traffic.json | tail -40
