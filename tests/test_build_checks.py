"""Build-time property of the register-resident kernels: the compiler must not touch the registers the inline asm
addresses by number (tools/check_regs.py).  The build itself runs the checker on the device assembly of every
kernel file it compiles (relate_amd/csrc/Makefile keeps that assembly as *.s and deletes an object that fails);
here the checker is run again over what the build left, and on a planted violation."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "relate_amd", "csrc")


def test_compiler_stays_out_of_reserved_registers():
    asm = sorted(glob.glob(os.path.join(CSRC, "paint_regs_g*.s")))
    if not asm:  # a tree that was built elsewhere (the assembly does not travel with it), or not built at all
        pytest.skip("no device assembly here: the build (make -C relate_amd/csrc) keeps it and checks it as it compiles")
    assert len(asm) == 4, "an incomplete build: four kernel files are compiled"
    for f in asm:
        obj = f[:-2] + ".o"
        assert os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(f) - 1, f"{obj} is older than its assembly"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_regs.py")] + asm, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "kernels clean" in p.stdout


def test_checker_flags_a_violation(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("_ZN6relate4regs14paint_kernel_rILi4ELi1EEEvNS0_5KArgsE:\n"
                   "\tv_add_f64 v[60:61], v[2:3], v[4:5]\n"
                   "\t;;#ASMSTART\n\tv_add_f64 v[70:71], v[70:71], v[2:3]\n\t;;#ASMEND\n"
                   "\t.end_amdhsa_kernel\n"
                   "_ZN6relate5aloop14paint_kernel_aILi4ELi1EEEvNS_4regs5KArgsE:\n"
                   "\ts_mov_b32 s30, s2\n"
                   "\tv_mov_b32 v20, v2\n"
                   "\t.end_amdhsa_kernel\n")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_regs.py"), str(bad)],
                       capture_output=True, text=True)
    assert p.returncode == 1 and "v[60:61]" in p.stderr and "s30" in p.stderr and "v20" not in p.stderr


def test_checker_reads_the_limits_a_kernel_states(tmp_path):
    """A kernel may state its own limits in its assembly (claim_registers / claim_slots): the compiler's scalar limit
    where it differs from the default, and the first accumulation register the asm addresses by number."""
    src = ("_ZN6relate4regs20repaint_bwd_kernel_rILi16ELi2EEEvNS0_12RepaintKArgsE:\n"
           "\t;;#ASMSTART\n\t; relate_limits v48 s44\n\t;;#ASMEND\n"
           "\t;;#ASMSTART\n\t; relate_limits a32\n\t;;#ASMEND\n"
           "\tv_accvgpr_write_b32 a31, v2\n"        # the compiler's own AGPRs: fine
           "\ts_mov_b32 s43, s2\n"
           "\t;;#ASMSTART\n\tv_accvgpr_write_b32 a[40], v[56]\n\t;;#ASMEND\n"
           "%s"
           "\t.end_amdhsa_kernel\n")
    ok, bad_a, bad_s = tmp_path / "ok.s", tmp_path / "bad_a.s", tmp_path / "bad_s.s"
    ok.write_text(src % "")
    bad_a.write_text(src % "\tv_accvgpr_read_b32 v3, a40\n")
    bad_s.write_text(src % "\ts_mov_b32 s44, s2\n")
    run = lambda f: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_regs.py"), str(f)],  # noqa: E731
                                   capture_output=True, text=True)
    assert run(ok).returncode == 0
    for f, what in ((bad_a, "a40"), (bad_s, "s44")):
        p = run(f)
        assert p.returncode == 1 and what in p.stderr


def test_checker_flags_a_scalar_load_at_a_loop_latch(tmp_path):
    """A compiler scalar load in a block that falls through into a loop header runs at every site and waits for the
    row fetch in flight; the same load in a conditional block of the loop body (a rare path) is fine."""
    head = "_ZN6relate4regs14paint_kernel_rILi20ELi4EEEvNS0_5KArgsE:\n"
    loop = (".LBB4_68:                               ; =>This Loop Header: Depth=1\n"
            "\ts_and_b32 s0, s33, 31\n"
            "\ts_cbranch_scc1 .LBB4_80\n"
            "; %%bb.69:\n"
            ".LBB4_70:                               ; %%rare\n"
            "                                        ;   in Loop: Header=BB4_68 Depth=1\n"
            "%s"
            "\ts_branch .LBB4_67\n"
            ".LBB4_80:\n"
            "                                        ;   in Loop: Header=BB4_68 Depth=1\n"
            "\ts_nop 0\n")
    latch = (".LBB4_67:                               ; %%Flow\n"
             "                                        ;   in Loop: Header=BB4_68 Depth=1\n"
             "%s"
             "\ts_add_i32 s0, s33, -1\n"
             "\ts_cbranch_scc1 .LBB4_104\n")
    load = "\ts_load_dwordx4 s[20:23], s[34:35], 0x58\n"
    tail = ".LBB4_104:\n\t.end_amdhsa_kernel\n"
    ok, bad = tmp_path / "ok.s", tmp_path / "bad.s"
    # (the latch precedes the header in the file, as the compiler lays the loop out)
    ok.write_text(head + latch % "" + loop % load + tail)
    bad.write_text(head + latch % load + loop % "" + tail)
    run = lambda f: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_regs.py"), str(f)],  # noqa: E731
                                   capture_output=True, text=True)
    assert run(ok).returncode == 0, run(ok).stderr
    p = run(bad)
    assert p.returncode == 1 and "loop latch" in p.stderr and "0x58" in p.stderr
