#!/bin/bash
# One parametrised runner for the GPU box (replaces the per-experiment job scripts of earlier rounds):
#   gpurun -- 'bash scripts/gpu_job.sh <tag> <step> [<step> ...]'      results -> gpurun_out/<tag>/
# steps:  tests[:<pytest args>]   pytest -m gpu (default: the whole suite)
#         bench[:<bench args>]    bench.py, the JSON line -> bench_line.json
#         py:<script and args>    python3 <script ...>    -> <script name>.txt
#         kstats:<driver args>    rocprofv3 --kernel-trace --stats of scripts/traffic_driver.py <args>
#         sh:<command>            anything else
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R" || exit 1
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p "$O"
export TMPDIR=/tmp
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  case $kind in
    tests) timeout 1500 python3 -m pytest ${arg:-tests} -m gpu -x -q 2>&1 | tail -15 | tee -a "$O/tests.txt" ;;
    bench) timeout 1200 python3 bench.py $arg 2> "$O/bench_stderr.txt" | tail -1 > "$O/bench_line.json"; tail -c 1500 "$O/bench_line.json"; echo ;;
    py) name=$(basename "${arg%% *}" .py); timeout 1200 python3 $arg > "$O/$name.txt" 2>&1; tail -40 "$O/$name.txt" ;;
    kstats) (cd /tmp && rm -rf /tmp/ks_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -- python3 "$R/scripts/traffic_driver.py" $arg > /dev/null 2>&1; python3 "$R/scripts/kstats.py" /tmp/ks_$TAG 30 > "$O/kstats_${arg// /_}.txt" 2>&1); tail -35 "$O/kstats_${arg// /_}.txt" ;;
    sh) bash -c "$arg" 2>&1 | tail -40 | tee -a "$O/sh.txt" ;;
    *) echo "unknown step $step" ;;
  esac
done
