#!/bin/bash
# round 5, first GPU job: new guards / tag format / bench sections
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_logging_sinks.py tests/test_gpu_rnn_config3.py tests/test_gpu_rnn_long.py tests/test_gpu_trainer.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r5a/pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5a/pytest.txt
timeout 300 ./scripts/handoff_lab/handoff 400000 > gpurun_out/r5a/handoff.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --trainer-utts 256 > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
echo "bench rc $?" >> gpurun_out/r5a/bench.err
tail -3 gpurun_out/r5a/pytest.txt; cat gpurun_out/r5a/handoff.txt | tail -8; tail -3 gpurun_out/r5a/bench.err
