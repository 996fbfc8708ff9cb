python -m pytest tests/test_gpu_mlpg.py tests/test_gpu_properties.py -q -m gpu -k mlpg 2>&1 | tail -3
for g in ${GEOMS:-16x16x4 16x8x4 8x16x4 8x8x4}; do echo "== $g"; ITTS_MLPG_GEOM=$g ITTS_MLPG_CHECK=1 python scripts/mlpg_curve.py fused 2>&1 | grep utts; done
MLPG_UTTS=4096 python scripts/mlpg_trace.py ${TRACE_GEOM:-8x16x4} | tail -16
