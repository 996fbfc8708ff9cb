"""The wave-per-transform FFT (csrc/wave_fft.h) against the workgroup-per-transform one it replaces in
the 1024-point frame kernels (wd::fft_lds / rfft_lds / irfft_lds): same butterflies, same twiddle
entries, same order of operations -- the outputs must be IDENTICAL bit for bit on random inputs
(complex forward / inverse, real forward / inverse, at 512 and at 1 024 complex points; 8 192 / 4 096 transforms each).  Replaces the FFTs inside
pyworld / pysptk that WorldFeatLabelGen.py:792-793, 940-943 and AudioProcessing.py:146-152, 252-255
reach.  The lab (scripts/wave_fft_lab/lab.hip) is built with hipcc on the spot; its schedule is also
checked as a computation graph, and for LDS bank conflicts, on the CPU (tests/test_wave_fft_schedule.py)."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave_fft_is_bit_identical_to_the_workgroup_fft(gpu, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "wave_fft_lab")
    src = os.path.join(ROOT, "scripts", "wave_fft_lab", "lab.hip")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-munsafe-fp-atomics",
                    "-Wno-unused-function", "-o", exe, src], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT, timeout=600)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    rows = re.findall(r"^(\S.*?)\s*: (\d+) of (\d+) values differ", res.stdout, flags=re.M)
    assert len(rows) == 8, res.stdout           # four transforms at 512 and at 1024 complex points
    for name, bad, total in rows:
        assert int(bad) == 0 and int(total) >= 4096 * 1024 * 2, (name, bad, total)
