"""Seeded synthetic chunks and the chunk-file writer (host-only entry points of the library)."""
import ctypes as C

import numpy as np

from ._lib import check, f64p, i32p, lib, u8p


def synth_chunk(N, L, seed, skewed_map=False):
    """(L,N) uint8 '0'/'1' matrix with linkage disequilibrium and r (L,) float64."""
    seq = np.empty((L, N), dtype=np.uint8)
    r = np.empty(L, dtype=np.float64)
    check(lib().relate_synth_chunk(N, L, seed, int(bool(skewed_map)), seq.ctypes.data_as(u8p), r.ctypes.data_as(f64p)))
    return seq, r


def window_boundaries(seq, memory_gb=5.0, max_windows=499):
    """Window boundaries by the memory rule of Data::MakeChunks (src/data.cpp:145,:219-255)."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    L, N = seq.shape
    wb = np.zeros(max_windows + 1, dtype=np.int32)
    nw = lib().relate_window_boundaries(N, L, seq.ctypes.data_as(u8p), float(memory_gb), max_windows,
                                        wb.ctypes.data_as(i32p))
    if nw < 0:
        check(nw)
    return wb[: nw + 1].copy()


def even_windows(L, W):
    """W windows of (nearly) equal length."""
    return np.round(np.linspace(0, L, W + 1)).astype(np.int32)


def write_chunk(output_dir, chunk_index, seq, r, wb):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    r = np.ascontiguousarray(r, dtype=np.float64)
    wb = np.ascontiguousarray(wb, dtype=np.int32)
    L, N = seq.shape
    check(lib().relate_write_chunk(str(output_dir).encode(), chunk_index, N, L, seq.ctypes.data_as(u8p),
                                   r.ctypes.data_as(f64p), len(wb) - 1, wb.ctypes.data_as(i32p)))
