"""ctypes access to the CPU oracle (oracle/libpaint_oracle.so) and, where it has
been built in this container, to the compiled reference translation unit
(oracle/_ref/libpaint_ref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def record_size(N):
    return 56 + 8 * N


class _PaintLib:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        f = getattr(self.lib, prefix + "_paint_stones")
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_int, _u8p, _f64p, C.c_double, C.c_double, C.c_int, _i32p, C.c_int, C.c_int, _u8p]
        self._paint = f
        g = getattr(self.lib, prefix + "_repaint_section")
        g.restype = C.c_int
        g.argtypes = [C.c_int, C.c_int, _u8p, _f64p, C.c_double, C.c_int, C.c_int, _f32p, _f32p, C.c_float,
                      C.c_float, C.c_int, _f32p, _f32p]
        self._repaint = g
        h = getattr(self.lib, prefix + "_fast_log")
        h.restype = C.c_float
        h.argtypes = [C.c_float]
        self.fast_log = h
        # the compiled reference keeps its Data between calls behind a handle (ref_data_*); the restatement
        # works on the caller's arrays directly and needs none
        self._handles = hasattr(self.lib, prefix + "_data_new")
        if self._handles:
            n = getattr(self.lib, prefix + "_data_new")
            n.restype = C.c_void_p
            n.argtypes = [C.c_int, C.c_int, _u8p, _f64p, C.c_double]
            fr = getattr(self.lib, prefix + "_data_free")
            fr.restype = None
            fr.argtypes = [C.c_void_p]
            ps = getattr(self.lib, prefix + "_data_paint_stones")
            ps.restype = C.c_int
            ps.argtypes = [C.c_void_p, C.c_double, C.c_int, _i32p, C.c_int, C.c_int, _u8p]
            self._new, self._free, self._paint_h = n, fr, ps

    def open_chunk(self, seq, r, theta_d):
        """A chunk kept by the library between paint_stones calls (no copy of the matrix per call)."""
        return _Chunk(self, seq, r, theta_d)

    def paint_stones(self, seq, r, theta_p, theta_d, wb, k_begin=0, k_end=None):
        """seq: (L,N) uint8 of '0'/'1'; returns (W, K, 56+8N) uint8."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        r = np.ascontiguousarray(r, dtype=np.float64)
        wb = np.ascontiguousarray(wb, dtype=np.int32)
        L, N = seq.shape
        W = len(wb) - 1
        k_end = N if k_end is None else k_end
        out = np.zeros((W, k_end - k_begin, record_size(N)), dtype=np.uint8)
        rc = self._paint(N, L, _ptr(seq, _u8p), _ptr(r, _f64p), theta_p, theta_d, W, _ptr(wb, _i32p), k_begin, k_end,
                         _ptr(out, _u8p))
        if rc != 0:
            raise RuntimeError(f"{self.prefix}_paint_stones failed: {rc}")
        return out

    def repaint_section(self, seq, r, theta, first, last, alpha_begin, beta_end, ls_a, ls_b, k):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        r = np.ascontiguousarray(r, dtype=np.float64)
        L, N = seq.shape
        cap = last - first + 1
        top = np.zeros((cap, N), dtype=np.float32)
        ls = np.zeros(cap, dtype=np.float32)
        ab = np.ascontiguousarray(alpha_begin, dtype=np.float32)
        be = np.ascontiguousarray(beta_end, dtype=np.float32)
        nd = self._repaint(N, L, _ptr(seq, _u8p), _ptr(r, _f64p), theta, first, last, _ptr(ab, _f32p), _ptr(be, _f32p),
                           ls_a, ls_b, k, _ptr(top, _f32p), _ptr(ls, _f32p))
        if nd < 0:
            raise RuntimeError("repaint_section failed")
        return top[:nd].copy(), ls[:nd].copy()


class _Chunk:
    def __init__(self, lib, seq, r, theta_d):
        self.lib = lib
        self.seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self.r = np.ascontiguousarray(r, dtype=np.float64)
        self.theta_d = theta_d
        self.L, self.N = self.seq.shape
        self.h = None
        if lib._handles:
            self.h = lib._new(self.N, self.L, _ptr(self.seq, _u8p), _ptr(self.r, _f64p), theta_d)

    def paint_stones(self, theta_p, wb, k_begin, k_end):
        wb = np.ascontiguousarray(wb, dtype=np.int32)
        W = len(wb) - 1
        out = np.zeros((W, k_end - k_begin, record_size(self.N)), dtype=np.uint8)
        if self.h is not None:
            rc = self.lib._paint_h(self.h, theta_p, W, _ptr(wb, _i32p), k_begin, k_end, _ptr(out, _u8p))
        else:
            rc = self.lib._paint(self.N, self.L, _ptr(self.seq, _u8p), _ptr(self.r, _f64p), theta_p, self.theta_d, W,
                                 _ptr(wb, _i32p), k_begin, k_end, _ptr(out, _u8p))
        if rc != 0:
            raise RuntimeError(f"paint_stones failed: {rc}")
        return out

    def close(self):
        if self.h is not None:
            self.lib._free(self.h)
            self.h = None

    __del__ = close


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "libpaint_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        _oracle = _PaintLib(path, "oracle")
        cd = _oracle.lib.oracle_count_derived
        cd.restype = C.c_int
        cd.argtypes = [C.c_int, C.c_int, _u8p, C.c_int, C.c_int, _i32p]
    return _oracle


def distance_matrix(top, ls, row, interp, wl, wr, N):
    """oracle_distance_matrix (restated DistanceMeasure::GetMatrix, parity unpinned): top is rows x pitch."""
    lib = oracle().lib
    f = lib.oracle_distance_matrix
    f.restype = None
    f.argtypes = [C.c_int, _f32p, C.c_int, _f32p, C.POINTER(C.c_int64), _u8p, _f64p, _f64p, _f32p]
    top = np.ascontiguousarray(top, dtype=np.float32)
    ls = np.ascontiguousarray(ls, dtype=np.float32)
    row = np.ascontiguousarray(row, dtype=np.int64)
    interp = np.ascontiguousarray(interp, dtype=np.uint8)
    wl = np.ascontiguousarray(wl, dtype=np.float64)
    wr = np.ascontiguousarray(wr, dtype=np.float64)
    out = np.zeros((N, N), dtype=np.float32)
    f(N, _ptr(top, _f32p), top.shape[1], _ptr(ls, _f32p), row.ctypes.data_as(C.POINTER(C.c_int64)), _ptr(interp, _u8p),
      _ptr(wl, _f64p), _ptr(wr, _f64p), _ptr(out, _f32p))
    return out


def count_derived(seq):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    L, N = seq.shape
    nd = np.zeros(N, dtype=np.int32)
    oracle().lib.oracle_count_derived(N, L, _ptr(seq, _u8p), 0, N, _ptr(nd, _i32p))
    return nd


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libpaint_ref.so"))


def ref():
    """The compiled reference (only exists where /root/reference was present at build time)."""
    global _ref
    if _ref is None:
        _ref = _PaintLib(os.path.join(ORACLE_DIR, "_ref", "libpaint_ref.so"), "ref")
    return _ref


def fast_build(kind):
    """The -O3 / AVX2 build of the compiled reference ("reference") or of the restatement ("port"), or None."""
    path = (os.path.join(ORACLE_DIR, "_ref", "libpaint_ref_o3.so") if kind == "reference"
            else os.path.join(ORACLE_DIR, "libpaint_oracle_o3.so"))
    if not os.path.exists(path):
        return None
    return _PaintLib(path, "ref" if kind == "reference" else "oracle")


# ---- record decoding -------------------------------------------------------
def decode_records(out):
    """(W,K,R) uint8 -> dict of arrays: the discrete fields and the float payloads."""
    W, K, R = out.shape
    N = (R - 56) // 8
    flat = out.reshape(W * K, R)

    def col(off, dt, n=1):
        v = flat[:, off:off + np.dtype(dt).itemsize * n].copy().view(dt)
        return v.reshape(W, K) if n == 1 else v.reshape(W, K, n)

    ob = 32 + 4 * N
    return {
        "start": col(0, np.int32), "end": col(4, np.int32),
        "a_rows": col(8, np.uint64), "a_cols": col(16, np.uint64), "a_snp": col(24, np.int32),
        "a_ls": col(28, np.float32), "alpha": col(32, np.float32, N),
        "b_rows": col(ob, np.uint64), "b_cols": col(ob + 8, np.uint64), "b_snp": col(ob + 16, np.int32),
        "b_ls": col(ob + 20, np.float32), "beta": col(ob + 24, np.float32, N),
    }
