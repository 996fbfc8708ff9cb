#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
cat $O/pytest.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
python3 - <<'PY'
import json
j=json.loads(open("gpurun_out/r5s/bench.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("metric","value","ms_per_step")}, j["roofline"]["frac"])
m=j.get("config",{})
for k in ("world","extra"):
    pass
def find(d, key):
    if isinstance(d, dict):
        if key in d: return d[key]
        for v in d.values():
            r=find(v,key)
            if r is not None: return r
    return None
ml=find(j,"mlpg"); print("mlpg", {k: ml[k] for k in ("ms","frac_of_hbm_peak")}, ml.get("batch_curve"))
PY
python3 scripts/mlpg_time.py 60 256
