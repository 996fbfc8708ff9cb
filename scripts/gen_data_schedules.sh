# gen_data lab: passes of 512 utterances on /dev/shm under different batch sizes / writer settings
cd $GRAFT_REPO_ROOT
run() { echo "== middle $1 chunk ${2:-32} writers ${3:-4} inflight ${4:-2} schedule '${5:-}'"; ITTS_GEN_DATA_INFLIGHT=${4:-2} ITTS_GEN_DATA_WRITE_CHUNK=${2:-32} ITTS_GEN_DATA_WRITERS=${3:-4} ITTS_GEN_DATA_SCHEDULE="${5:-}" timeout 300 python3 scripts/prof_gen_data.py 512 $1 --dir=/dev/shm 2>&1 | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read())['gen_data']; print('  passes', [round(t*1e3,1) for t in j['all_passes']], 'rtf %.3g' % j['rtf'])"; }
run 64
run 64 16 6
run 64 64 2
run 128
run 128 16 8
run 96 32 4 2 "32"
run 128 32 4 2 "32,64"
