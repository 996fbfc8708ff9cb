"""The DPP products of mcls_solve_dpp_kernel are inline asm (v_fmac_f64_dpp ... row_newbcast): the compiler's
hazard recognizer does not look inside asm statements, and on gfx9 a VALU write of a VGPR needs two wait
states before a DPP read of it.  The kernel puts an s_nop 1 in front of every step's sequence and feeds the DPP
operand from LDS loads; this script compiles csrc/mcep_lockstep.hip to ISA and checks every DPP product: none
of the two instructions in front of it may be a VALU write of its DPP source registers.
usage: python scripts/dpp_hazard_scan.py      (exit code 1 and the offending pairs on a hazard)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-ffp-contract=on",
         "-munsafe-fp-atomics"]


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(asm_text):
    lines = [l.strip() for l in asm_text.splitlines()]
    lines = [l for l in lines if l and not l.startswith(";") and not l.startswith(".") and not l.endswith(":")]
    checked, bad = 0, []
    for i, l in enumerate(lines):
        if not l.startswith("v_fmac_f64_dpp"):
            continue
        checked += 1
        src = regs(l.split(None, 1)[1].split(",")[1].strip())
        for p in lines[max(0, i - 2):i]:
            if p.startswith("v_") and not p.startswith("v_cmp") and not p.startswith("v_fmac_f64_dpp"):
                if regs(p.split(None, 1)[1].split(",")[0].strip()) & src:
                    bad.append((p, l))
    return checked, bad


def main():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "idiaptts_amd", "csrc", "mcep_lockstep.hip")
    with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "idiaptts_amd", "_lib") if os.path.isdir(
            os.path.join(ROOT, "idiaptts_amd", "_lib")) else None) as tmp:
        out = os.path.join(tmp, "mcep_lockstep.s")
        subprocess.run([hipcc] + FLAGS + ["-S", "--cuda-device-only", src, "-o", out], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        checked, bad = scan(open(out).read())
    print("DPP products checked: %d, hazards: %d" % (checked, len(bad)))
    for p, l in bad[:10]:
        print("  ", p, "->", l)
    return 1 if bad or checked == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
