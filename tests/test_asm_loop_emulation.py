"""The generated assembly loop of the painting kernel (tools/gen_paint_loop.py) interpreted on the CPU
(tools/emu_paint_loop.py): every scalar load, row touch and LDS access of the loop inside its buffer, no
register read before its load has been waited for, and the stones equal to the reference's records bit for bit.
A memory fault of that loop on the GPU costs a machine; here it is an exception."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import emu_paint_loop as emu  # noqa: E402
import oracle_lib as ol  # noqa: E402
import synth_np as sy  # noqa: E402
from conftest import load_golden  # noqa: E402


def test_first_and_last_targets_of_a_golden_chunk():
    """Target 0's backward pass ends at the first record of the record buffer, the last target's forward pass at
    the last one: the two places where a fetch one record too far leaves the allocation."""
    g = load_golden("n130_tail")
    bad, n, _ = emu.check_case(g["seq"], g["r"], g["wb"], g["theta_painter"], g["stones"], 4, 1, [0, 64, 129], verbose=False)
    assert bad == 0 and n > 0


def test_toy_chunk_all_targets():
    g = load_golden("toy_r001")
    bad, n, _ = emu.check_case(g["seq"], g["r"], g["wb"], g["theta_painter"], g["stones"], 4, 1, list(range(g["N"])), verbose=False)
    assert bad == 0


@pytest.mark.parametrize("N,G,MPL", [(600, 2, 8), (600, 4, 4), (1100, 1, 20)])
def test_several_waves_and_several_rounds(N, G, MPL):
    """Workgroups of several waves (the exchange through LDS, the barrier) and lanes of more than 16 words (two
    rounds of row words per site), against the oracle."""
    seq, r, wb = sy.make_chunk(N=N, L=160, W=2, seed=N + G, r_mode="skew")
    want = ol.oracle().paint_stones(seq, r, 0.001, 0.001, wb)
    bad, n, stats = emu.check_case(seq, r, wb, 0.001, want, MPL, G, [0, N - 1], verbose=False)
    assert bad == 0
    assert stats["valu"] / stats["wave_sites"] < 3 * MPL + 36


def test_a_fetch_past_the_list_is_caught():
    """The checker itself: the loop as it was first written fetched one record beyond each list, 24 bytes before
    the record buffer in target 0's backward pass."""
    g = load_golden("toy_r001")
    body = emu.gen.loop_body(4, 1)
    unguarded = [ln for ln in body if "Lnorec" not in ln]
    i = next(i for i, ln in enumerate(unguarded) if ln.startswith("s_load_dwordx4"))
    assert unguarded[i - 1].startswith("s_cmp_eq_u32")  # the guard's compare stays, harmless without its branch
    ch = emu.Chunk(g["seq"], g["r"], g["wb"], g["theta_painter"])
    labels = emu.index_labels([ln.replace("%=", "") for ln in unguarded])
    with pytest.raises(emu.EmuError, match="outside every buffer"):
        emu.paint_target_pass(ch, 0, 1, 4, 1, unguarded, labels)
