"""Phased haplotypes (.haps / .sample / genetic map) as bit rows, cut into the chunks and windows of
Data::MakeChunks (src/data.cpp:134-528).  Host-side mirror of section 6 of include/relate_gpu.h; parsing and
cutting need no GPU, loading a chunk into a Plan does."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check, lib


class Haps:
    def __init__(self, haps_path, sample_path, map_path, memory_gb=5.0):
        self._h = C.c_void_p()
        check(lib().relate_haps_open(os.fsencode(haps_path), os.fsencode(sample_path), os.fsencode(map_path),
                                     float(memory_gb), C.byref(self._h)))
        n, l, c = C.c_int(), C.c_int(), C.c_int()
        check(lib().relate_haps_dims(self._h, C.byref(n), C.byref(l), C.byref(c)))
        self.N, self.L, self.num_chunks = n.value, l.value, c.value

    def close(self):
        if self._h:
            lib().relate_haps_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def chunk(self, c):
        """(first_snp, L_chunk, window boundaries relative to first_snp) of chunk c."""
        first, lc, nw = C.c_int(), C.c_int(), C.c_int()
        check(lib().relate_haps_chunk_dims(self._h, int(c), C.byref(first), C.byref(lc), C.byref(nw)))
        wb = np.empty(nw.value + 1, dtype=np.int32)
        check(lib().relate_haps_chunk_windows(self._h, int(c), wb.ctypes.data_as(_lib.i32p)))
        return first.value, lc.value, wb

    def positions(self):
        """bp (L+1), dist (L), rpos (L+1), r (L) of the whole data set."""
        bp = np.empty(self.L + 1, dtype=np.int32)
        dist = np.empty(self.L, dtype=np.int32)
        rpos = np.empty(self.L + 1, dtype=np.float64)
        r = np.empty(self.L, dtype=np.float64)
        check(lib().relate_haps_positions(self._h, bp.ctypes.data_as(_lib.i32p), dist.ctypes.data_as(_lib.i32p),
                                          rpos.ctypes.data_as(_lib.f64p), r.ctypes.data_as(_lib.f64p)))
        return bp, dist, rpos, r

    def rows(self):
        """The bit matrix, (L, row_stride_words) uint64: a copy."""
        ptr, stride = C.POINTER(C.c_uint64)(), C.c_int()
        check(lib().relate_haps_rows(self._h, C.byref(ptr), C.byref(stride)))
        return np.ctypeslib.as_array(ptr, shape=(self.L, stride.value)).copy()

    def write_chunks(self, output_dir, use_transitions=True):
        check(lib().relate_haps_write_chunks(self._h, os.fsencode(output_dir), 1 if use_transitions else 0))

    def load_chunk(self, plan, c, rho=1.0):
        """Chunk c into a relate_amd.Plan: bit rows and r * rho to the device, transpose built there."""
        check(lib().relate_plan_load_haps_chunk(plan._h, self._h, int(c), float(rho)))
        first, lc, wb = self.chunk(c)
        plan._set(self.N, lc, wb)


def row_stride_words(N):
    return lib().relate_row_stride_words(int(N))
