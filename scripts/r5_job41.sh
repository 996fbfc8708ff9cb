#!/bin/bash
# evidence run at the head of the branch (after the phase scan's sequential first round and the two events of
# itts_world_synthesize_after): full GPU suite, the driver's bench protocol, smoke, the synthesis timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r5ak}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest.txt
cat $O/pytest.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench.err
echo "bench rc $?"
python3 -c "
import json
j=json.loads(open('$O/bench_20_5.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['world']['analysis_ms'], j['world']['synthesis_ms'], j['world_48k']['analysis_ms'], j['world_48k']['synthesis_ms'], j['mlpg']['ms'], j['mlpg']['ms_queued'], j['mlpg']['roofline']['frac'], [(c['utterances'], round(c['ms'],3), round(c['frac_of_hbm_peak'],3), round(c['frac_of_hbm_peak_queued'],3)) for c in j['mlpg']['batch_curve']])
print(j.get('bilstm',{}).get('ms_per_step'), j.get('bigru',{}).get('ms_per_step'))
"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash scripts/syn_timeline.sh > $O/synthesis_timeline.txt 2>&1; tail -25 $O/synthesis_timeline.txt
