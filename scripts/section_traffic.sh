# HBM traffic per pass / step of the bench's secondary sections (run on the GPU box):
#   bash scripts/section_traffic.sh <tag>   ->  gpurun_out/<tag>_section_traffic.json
# Two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) at 1 and at 3 passes of every
# section; bytes per pass = (counters at 3 - counters at 1) / 2, so set-up and warm-up cancel.
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st; mkdir -p /tmp/st
for spec in "analysis 16000" "synthesis 16000" "analysis 48000" "synthesis 48000" "bilstm 0" "bigru 0" "mlpg 0"; do
  set -- $spec; sec=$1; fs=$2
  for n in 3 5; do
    for c in FETCH_SIZE WRITE_SIZE; do
      d=/tmp/st/${sec}_${fs}_${n}_${c}
      rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/scripts/traffic_driver.py $sec $n $fs > $d.log 2>&1
    done
  done
done
python3 $R/scripts/traffic_summary.py /tmp/st $O/${TAG}_section_traffic.json
