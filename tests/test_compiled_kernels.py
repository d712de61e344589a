"""The painting kernels as COMPILED -- the gfx950 assembly the build keeps of every register-resident kernel file
(relate_amd/csrc/paint_regs_g*.s: compiler-generated code and the hand-written instruction streams inside it) --
interpreted on the CPU (tools/gfx950_emu.py) against the reference's records: every global, scalar, flat, scratch
and LDS access inside its buffer, no scalar register used before its load has been waited for, every stored float
the reference's.  This is what stands in for a GPU where there is none; the GPU tests say the same of the hardware."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle_lib as ol  # noqa: E402
import synth_np as sy  # noqa: E402
from conftest import load_golden  # noqa: E402

CSRC = os.path.join(ROOT, "relate_amd", "csrc")
pytestmark = pytest.mark.skipif(len(glob.glob(os.path.join(CSRC, "paint_regs_g*.s"))) < 4,
                                reason="no device assembly here: the build (make -C relate_amd/csrc) keeps it")


def same_records(got, want):
    g, w = ol.decode_records(got), ol.decode_records(want)
    for key in g:
        a, b = g[key], w[key]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), key


@pytest.mark.parametrize("name,targets", [("n130_tail", (0, 2)), ("n130_tail", (128, 130)), ("n70_skew_emptywin", (63, 66)),
                                          ("toy_r001", (0, 5))])
def test_one_wave_kernel_on_golden_chunks(name, targets):
    import emu_kernel as ek
    g = load_golden(name)
    t = range(*targets)
    out, stats = ek.paint_compiled(g["seq"], g["r"], g["wb"], g["theta_painter"], g["theta_data"], t, 1, 4)
    same_records(out, g["stones"][:, targets[0]:targets[1]])
    assert sum(stats.values()) > 1000


def test_two_theta_kernel_on_its_golden_chunk():
    """FastPainting(N, theta) beside Data::theta (the Rust caller): the two-row backward pass, which also executes
    the compiler's scratch spills."""
    import emu_kernel as ek
    g = load_golden("n33_mixed_theta")
    assert g["theta_painter"] != g["theta_data"]
    out, _ = ek.paint_compiled(g["seq"], g["r"], g["wb"], g["theta_painter"], g["theta_data"], range(30, 33), 1, 4)
    same_records(out, g["stones"][:, 30:33])


@pytest.mark.parametrize("N,G,MPL,tp,td", [(600, 2, 8, 0.001, 0.001), (600, 4, 4, 0.001, 0.001), (1100, 1, 20, 0.001, 0.001),
                                           (600, 2, 8, 0.001, 0.01), (4000, 8, 8, 0.001, 0.001)])
def test_several_waves_and_several_rounds(N, G, MPL, tp, td):
    """Workgroups of several waves (LDS exchange, barrier per site), lanes of more than 16 words (two rounds of row
    words per site), one and two thetas, against the oracle; targets at both ends of the panel and across a word."""
    import emu_kernel as ek
    seq, r, wb = sy.make_chunk(N=N, L=120, W=2, seed=N + G, r_mode="skew")
    for k0 in (0, 63, N - 2):
        want = ol.oracle().paint_stones(seq, r, tp, td, wb, k0, k0 + 2)
        out, _ = ek.paint_compiled(seq, r, wb, tp, td, range(k0, k0 + 2), G, MPL)
        same_records(out, want)
