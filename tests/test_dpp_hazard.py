"""mcls_solve_dpp_kernel's products are inline asm with a DPP operand (csrc/mcep_lockstep.hip): the hazard the
compiler would otherwise handle -- a VALU write of a register within two instructions in front of a DPP read of
it -- is checked on the generated code itself (scripts/dpp_hazard_scan.py; hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scanner_sees_a_hazard_when_there_is_one():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dpp_hazard_scan as d
    ok = "ds_read_b64 v[0:1], v9\nv_mul_f64 v[4:5], v[6:7], v[8:9]\ns_nop 1\nv_fmac_f64_dpp v[10:11], v[0:1], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
    bad = "v_mov_b64_e32 v[0:1], v[20:21]\nv_fmac_f64_dpp v[10:11], v[0:1], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
    assert d.scan(ok) == (1, [])
    n, hz = d.scan(bad)
    assert n == 1 and len(hz) == 1


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_generated_code_has_no_dpp_hazard():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dpp_hazard_scan.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-2000:]
    assert "hazards: 0" in res.stdout and "checked: 0" not in res.stdout
