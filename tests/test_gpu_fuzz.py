"""Random-shape sweeps (scripts/mlpg_fuzz.py, scripts/gemm_fuzz.py) as tests: every MLPG path against
the sequential sweeps, the fp32 GEMM entry points against torch in float64.  Child processes: the
MLPG script switches ITTS_MLPG_PATH per call, which the library reads from the environment."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,cases,seed", [("mlpg_fuzz.py", 80, 5), ("gemm_fuzz.py", 60, 9)])
def test_random_shapes(gpu, script, cases, seed):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(cases), str(seed)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "cases %d" % cases in res.stdout
