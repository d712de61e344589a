"""Small seeded synthetic chunks for the CPU-side tests and the golden fixtures
(numpy only; the large benchmark inputs come from the library's own generator).

A chunk is: seq (L,N) uint8 of '0'/'1' (row = SNP, as chunk_<c>.hap holds it),
r (L,) float64 (already multiplied by rho) and window boundaries wb (W+1,) int32
with wb[0] == 0 and wb[W] == L."""
import numpy as np


def make_chunk(N, L, W, seed, r_mode="flat", singleton_frac=0.3, empty_window=False):
    rng = np.random.default_rng(seed)
    perm = rng.permutation(N)
    seq = np.full((L, N), ord("0"), dtype=np.uint8)
    for s in range(L):
        if s % 37 == 36:  # drift the ordering: LD decays along the chunk
            i, j = rng.integers(0, N, 2)
            v = perm[i]
            perm = np.delete(perm, i)
            perm = np.insert(perm, j % (N - 1), v)
        if rng.random() < singleton_frac:
            c = 1
        else:
            # ~1/i site-frequency spectrum
            c = int(min(N - 1, max(1, np.floor(np.exp(rng.random() * np.log(N - 1))))))
        a = rng.integers(0, N - c + 1)
        seq[s, perm[a:a + c]] = ord("1")
    if r_mode == "flat":
        r = np.full(L, 0.025) * rng.uniform(0.5, 1.5, L)
    elif r_mode == "zero":
        r = np.zeros(L)
    elif r_mode == "skew":  # hotspots: spans the 0.99 clamp and both rescale directions
        r = np.exp(rng.uniform(np.log(2.5e-7), np.log(10.0), L))
    else:
        raise ValueError(r_mode)
    # window edges: random increasing, last == L
    inner = np.sort(rng.choice(np.arange(1, L), size=W - 1, replace=False)) if W > 1 else np.array([], dtype=int)
    wb = np.concatenate([[0], inner, [L]]).astype(np.int32)
    if empty_window and W >= 3:
        # make one window so narrow that most targets have no derived site in it
        wb[2] = wb[1] + 1
        wb = np.sort(wb).astype(np.int32)
    return seq, r.astype(np.float64), wb


def toy_test_painting():
    """The N=5, L=10 haplotypes and the expected mismatch-count matrix of the
    reference's own known-answer test (test/test_painting.cpp:17-82)."""
    cols = ["0110000000", "0110010100", "0100000000", "0000100000", "0000100000"]
    seq = np.array([[ord(cols[n][s]) for n in range(5)] for s in range(10)], dtype=np.uint8)
    d = np.array([[0, 0, 1, 2, 2], [2, 0, 3, 4, 4], [0, 0, 0, 1, 1], [1, 1, 1, 0, 0], [1, 1, 1, 0, 0]],
                 dtype=np.float32)
    return seq, d
