#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_world.py tests/test_gpu_properties.py tests/test_gpu_dropin.py tests/test_gpu_mgc.py -m gpu -x -q 2>&1 | tail -2
python3 - <<'PY'
import os, json, torch, bench
dev = torch.device("cuda", 0)
for fs, n in ((16000, 256), (48000, 64)):
    for side in ("1", "0"):
        os.environ["ITTS_BENCH_SYNTH_SIDE"] = side
        r = bench.world_section(dev, n, fs, with_cpu=False, with_mlpg=False, key="w")
        w = r["w"]
        print(fs, "side stream" if side == "1" else "one stream ", "synthesis %.2f ms  analysis %.2f ms" % (w["synthesis_ms"], w["analysis_ms"]))
PY
