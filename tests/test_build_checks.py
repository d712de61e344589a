"""Build-time property of the register-resident kernels: the compiler must not touch the registers the
inline asm addresses by number (tools/check_regs.py).  Compiles one instantiation file to assembly (about
half a minute) and runs the checker on it; `make -C relate_amd/csrc check-regs` covers all four."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "relate_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.timeout(600)
def test_compiler_stays_out_of_reserved_registers(tmp_path):
    asm = tmp_path / "paint_regs_g1.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                    "--cuda-device-only", os.path.join(CSRC, "paint_regs_g1.hip"), "-o", str(asm)],
                   check=True, capture_output=True)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_regs.py"), str(asm)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "kernels clean" in p.stdout


def test_checker_flags_a_violation(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("_ZN6relate4regs14paint_kernel_rILi4ELi1EEEvNS0_5KArgsE:\n"
                   "\tv_add_f64 v[60:61], v[2:3], v[4:5]\n"
                   "\t;;#ASMSTART\n\tv_add_f64 v[70:71], v[70:71], v[2:3]\n\t;;#ASMEND\n"
                   "\t.end_amdhsa_kernel\n")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_regs.py"), str(bad)],
                       capture_output=True, text=True)
    assert p.returncode == 1 and "v[60:61]" in p.stderr
