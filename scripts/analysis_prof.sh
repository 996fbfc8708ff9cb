# Per-kernel time of the WORLD analysis alone (GPU box): bash scripts/analysis_prof.sh <tag> [utts] [fs]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ap && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ap -- python3 $R/scripts/world_analysis_only.py ${2:-256} ${3:-16000} 4 > /tmp/ap.log 2>&1
tail -3 /tmp/ap.log
python3 $R/scripts/kstats.py /tmp/ap 24 | tee $O/$1_analysis_kstats_${3:-16000}.txt
