"""The sum of a set without its R largest (csrc/select_largest.h: per-wave tournaments by DPP, ranks by
binary search over the four waves' lists, batches of R / 3 rounds) against a sort on the host.  This is
the selection inside D4C's coarse aperiodicity (WORLD d4c.cpp sorts each band's power spectrum;
reached from WorldFeatLabelGen.py:792-805 through pyworld.d4c).  Real spectra hardly ever hold equal
values, so the cases that decide the tie handling -- a few distinct values, one value, fewer distinct
values than R -- come from scripts/select_lab/lab.hip, built with hipcc on the spot."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rest_without_the_largest_matches_a_sort(gpu, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "select_lab")
    src = os.path.join(ROOT, "scripts", "select_lab", "lab.hip")
    inc = os.path.join(ROOT, "idiaptts_amd", "csrc")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + inc, "-o", exe, src], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    rows = re.findall(r"^n\s+(\d+): (\d+) sets x 15 values of R x 4 mixes, (\d+) wrong", res.stdout, flags=re.M)
    assert len(rows) == 3 and all(int(bad) == 0 for _, _, bad in rows), res.stdout
    assert "all sums agree" in res.stdout
