#!/bin/bash
# the offsets kernels by a wave instead of a thread: synthesis timeline + the tests that synthesise
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/${TAG:-r5at}; mkdir -p $O
bash scripts/syn_timeline.sh > $O/synthesis_timeline.txt 2>&1; tail -17 $O/synthesis_timeline.txt
timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_loader.py tests/test_gpu_dropin.py tests/test_gpu_properties.py tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
timeout 600 python bench.py --steps 5 --warmup 2 --ramp-steps 0 --no-cpu-baseline --bilstm-utts 0 --trainer-utts 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench: 16k analysis %.3f synthesis %.3f ms  48k analysis %.3f synthesis %.3f ms' % (j['world']['analysis_ms'], j['world']['synthesis_ms'], j['world_48k']['analysis_ms'], j['world_48k']['synthesis_ms']))" | tee $O/bench_world.txt
