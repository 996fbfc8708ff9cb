#!/bin/bash
# (1) two ranks on one device (functional check of bench.py's N > 1 path at the head of the branch: gloo collectives)
# (2) how much of the pulse kernels is their overlap-add's atomics?  (SYN_WAVE_DIAG=1: plain conditional stores instead)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5al; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --share-gpu --world-utts 16 --bilstm-utts 8 --trainer-utts 0 > $O/bench_2ranks_shared.json 2> $O/bench_2ranks.err
echo "2-rank bench rc $?"; tail -c 600 $O/bench_2ranks_shared.json | cut -c1-600; tail -3 $O/bench_2ranks.err
L=idiaptts_amd/_lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for def in "-DSYN_WAVE_DIAG=1" ""; do
  /opt/rocm/bin/hipcc $FLAGS $def -c idiaptts_amd/csrc/synth.hip -o $L/synth.o 2>/dev/null || exit 2
  /opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $L/*.o || exit 3
  echo "== ${def:-atomics}" | tee -a $O/atomics_ab.txt
  bash scripts/syn_timeline.sh 2>&1 | grep -E "pulse_wave|synthesis:" | tee -a $O/atomics_ab.txt
done
