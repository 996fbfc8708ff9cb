# Same-box A/B of the FF step: the library as built against one whose nn.o comes from another commit.
# Prepare here (no GPU needed):
#   mkdir -p /tmp/oldsrc && git archive <commit> idiaptts_amd/csrc include | tar -x -C /tmp/oldsrc
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -munsafe-fp-atomics \
#         -c /tmp/oldsrc/idiaptts_amd/csrc/nn.hip -o scripts/ff_ab/nn_old.o      (*.o is git-ignored, it travels with gpurun)
# then: gpurun -- 'bash scripts/ff_ab/run.sh'.  Used in round 4 for the grouped tile order: as a run-time branch in
# decode_tile it cost the FF step 1 % (1.0184 against 1.0087 ms, three alternating pairs); as a kernel of its own, nothing
# (1.0041 / 1.0043 / 1.0081 against 1.0047 / 1.0085 / 1.0088).
R=$GRAFT_REPO_ROOT; L=$R/idiaptts_amd/_lib; cd /tmp
FF="--steps 20 --warmup 5 --world-utts 0 --bilstm-utts 0 --no-cpu-baseline"
line() { python3 $R/bench.py $FF 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['roofline']['gemm_ms_per_step'],4))"; }
cp $L/libidiaptts_amd.so /tmp/lib_new.so
OBJS=$(ls $L/*.o | grep -v "/nn.o")
hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o /tmp/lib_old.so $OBJS $R/scripts/ff_ab/nn_old.o
for rep in 1 2 3; do
  cp /tmp/lib_new.so $L/libidiaptts_amd.so; line new
  cp /tmp/lib_old.so $L/libidiaptts_amd.so; line old
done
cp /tmp/lib_new.so $L/libidiaptts_amd.so
