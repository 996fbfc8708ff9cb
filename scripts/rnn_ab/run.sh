# Same-box A/B of the forward recurrence's exchange format (GPU box): the library as built against one
# built with the round-3 / early round-4 header (pairs P, P ^ mask(step)).  bash scripts/rnn_ab/run.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=$R/idiaptts_amd/_lib
cp $L/libidiaptts_amd.so /tmp/lib_orig.so
rm -rf /tmp/old_tree; mkdir -p /tmp/old_tree/idiaptts_amd; cp -r $R/idiaptts_amd/csrc /tmp/old_tree/idiaptts_amd/csrc; cp -r $R/include /tmp/old_tree/include
cp $R/scripts/rnn_ab/rnn_persist_pairs.h /tmp/old_tree/idiaptts_amd/csrc/rnn_persist.h
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
hipcc $FLAGS -c /tmp/old_tree/idiaptts_amd/csrc/lstm.hip -o /tmp/lstm_old.o || exit 1
hipcc $FLAGS -c /tmp/old_tree/idiaptts_amd/csrc/gru.hip -o /tmp/gru_old.o || exit 1
OBJS=$(ls $L/*.o | grep -v "/lstm.o\|/gru.o")
run() { python3 $R/bench.py --steps 5 --warmup 2 --world-utts 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'bilstm %.2f ms  bigru %.2f ms  ff %.4f ms' % (d['bilstm']['ms_per_step'], d['bigru']['ms_per_step'], d['ms_per_step']))"; }
for rep in 1 2; do
  cp /tmp/lib_orig.so $L/libidiaptts_amd.so; run tagged | tee -a $O/$1_rnn_ab.txt
  hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $OBJS /tmp/lstm_old.o /tmp/gru_old.o || exit 1
  run pairs | tee -a $O/$1_rnn_ab.txt
done
cp /tmp/lib_orig.so $L/libidiaptts_amd.so
