"""CPU check of the wave FFT's schedule (scripts/wave_fft_sim.py): the three register passes and two LDS
exchanges compute exactly the butterflies of the radix-2 DIT transform they replace (values tracked
as hashes of their computation tree: same operands, same twiddle-table entries, same order), and no
LDS access of the schedule has a bank conflict under the lane groups of MI355X_MICROARCH.md."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedule_equals_radix2_dit_and_is_conflict_free():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "wave_fft_sim.py")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout
    assert res.stdout.count("same computation DAG as fft_lds: True") == 2        # 512 and 1024 points
    assert res.stdout.count("bank conflicts: 0") == 2
