#!/bin/bash
# the noise generator on the library's side stream (beside the phase scan) against inline (ITTS_SYNTH_NOISE_INLINE=1)
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5ap; mkdir -p $O
for inl in 1 ""; do
  if [ -n "$inl" ]; then export ITTS_SYNTH_NOISE_INLINE=1; else unset ITTS_SYNTH_NOISE_INLINE; fi
  echo "== inline=${inl:-0}" | tee -a $O/noise_side_ab.txt
  bash scripts/syn_timeline.sh 2>&1 | grep -E "randn|pulse_wave|synthesis:|syn_cast" | tee -a $O/noise_side_ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --ramp-steps 0 --no-cpu-baseline --bilstm-utts 0 --trainer-utts 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench: 16k synthesis %.3f ms  48k synthesis %.3f ms' % (j['world']['synthesis_ms'], j['world_48k']['synthesis_ms']))" | tee -a $O/noise_side_ab.txt
done
unset ITTS_SYNTH_NOISE_INLINE
timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_loader.py tests/test_gpu_dropin.py tests/test_gpu_properties.py tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
