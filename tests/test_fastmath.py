"""csrc/fastmath.h (the short fp64 log / sincos / cos / exp of the WORLD per-bin loops) on the host:
the header compiles for both sides, so its accuracy is checked here against long double libm over
the domains the kernels use -- the oracle's libm calls are correctly rounded to < 1 ulp, and so must
these be (replaces pyworld's / pysptk's libm calls inside CheapTrick, mcep and Synthesis:
WorldFeatLabelGen.py:792-793, 940-943)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fastmath_is_accurate_to_one_ulp(tmp_path):
    exe = str(tmp_path / "fastmath_check")
    src = os.path.join(ROOT, "tests", "native", "fastmath_check.cpp")
    for flags in (["-ffp-contract=off"], ["-mfma", "-ffp-contract=fast"]):   # as written / contracted
        subprocess.run(["g++", "-O2", "-std=c++17"] + flags + ["-o", exe, src], check=True)
        out = json.loads(subprocess.run([exe], check=True, stdout=subprocess.PIPE, text=True).stdout)
        assert out["exact"], out
        for k in ("exp", "log", "sin", "cos", "fcos", "fsin"):
            assert out[k] < 1.0, (flags, out)
        assert out["sin_abs"] < 1.2e-16 and out["cos_abs"] < 1.2e-16, out
