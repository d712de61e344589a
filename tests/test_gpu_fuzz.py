"""Seeded random configurations of the painting on the GPU against the CPU oracle (run with -m gpu).

The fixed cases of test_gpu_parity.py pin the shapes the reference's own tests use; this sweep walks
the corners between them: N around multiples of 64, very short lists, one window, sub-ranges of
targets, one theta and two, every wave count, flat and skewed recombination maps.  Same bar:
discrete fields bit-exact, floats within 1e-5 relative -- and, as measured on every case here, bit-identical."""
import numpy as np
import pytest

import oracle_lib as ol
import synth_np as sy
from test_gpu_parity import compare_records, gpu_paint

pytestmark = pytest.mark.gpu


def _cases(n_cases, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_cases):
        kind = i % 4
        if kind == 0:
            N = int(rng.integers(2, 130))
        elif kind == 1:
            N = int(64 * rng.integers(1, 12) + rng.integers(-1, 2))  # 64m-1, 64m, 64m+1
        elif kind == 2:
            N = int(rng.integers(130, 900))
        else:
            N = int(rng.integers(257, 700))
        N = max(N, 2)
        L = int(rng.integers(12, 500))
        W = int(rng.integers(1, min(6, L - 1) + 1))
        mode = ("flat", "skew", "zero")[int(rng.integers(0, 3))]
        tp = float(rng.choice([1e-4, 0.001, 0.01, 0.025]))
        td = tp if rng.random() < 0.6 else float(rng.choice([0.001, 0.01, 0.05]))
        waves = int(rng.choice([0, 0, 2, 4, 8]))
        sub = rng.random() < 0.3 and N > 8
        k0 = int(rng.integers(0, N // 2)) if sub else 0
        k1 = int(rng.integers(k0 + 1, N + 1)) if sub else N
        out.append((N, L, W, 1000 + i, mode, tp, td, waves, k0, k1))
    return out


@pytest.mark.parametrize("N,L,W,seed,mode,tp,td,waves,k0,k1", _cases(48, 7))
def test_random_configuration(N, L, W, seed, mode, tp, td, waves, k0, k1):
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=seed, r_mode=mode, empty_window=(W >= 3 and seed % 2 == 0))
    want = ol.oracle().paint_stones(seq, r, tp, td, wb, k0, k1)
    got = gpu_paint(seq, r, tp, td, wb, 1, waves, k0, k1)
    frac = compare_records(got, want, N, f"N={N} L={L} W={W} {mode} thetas {tp},{td} waves {waves} k [{k0},{k1})")
    assert frac == 0.0
