"""ctypes access to the tree-builder oracle (oracle/treebuilder_oracle.cpp in libpaint_oracle.so) and to the scan
kernels compiled for the host (tests/hostrun/).  Test infrastructure only."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_f32p, _i32p, _f64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)


def fixtures():
    return json.load(open(os.path.join(HERE, "golden", "treebuilder_fixtures.json")))


def _lib():
    lib = ol.oracle().lib
    lib.oracle_minmatch_threshold.restype = C.c_float
    lib.oracle_minmatch_threshold.argtypes = [C.c_double]
    lib.oracle_minmatch_scan.restype = C.c_longlong
    lib.oracle_minmatch_scan.argtypes = [C.c_int, _f32p, C.c_float, _f32p, C.c_longlong, _i32p, _i32p, _f32p]
    lib.oracle_quick_build.restype = C.c_int
    lib.oracle_quick_build.argtypes = [C.c_int, _f32p, C.c_double, C.c_int, _f64p, _i32p]
    lib.oracle_minmatch_initialize.restype = None
    lib.oracle_minmatch_initialize.argtypes = [C.c_int, _f32p, C.c_double, C.c_int, _f64p, _i32p, _i32p, _f64p, _f64p, _i32p,
                                               _f64p, _f32p]
    lib.oracle_uniform_draws.restype = None
    lib.oracle_uniform_draws.argtypes = [C.c_uint, C.c_int, _f64p]
    return lib


def threshold(theta):
    return float(_lib().oracle_minmatch_threshold(theta))


def _scan(fn, d, thr):
    d = np.ascontiguousarray(d, dtype=np.float32)
    N = d.shape[0]
    mv = np.zeros(N, dtype=np.float32)
    cap = N * (N - 1) // 2 + 1
    ci, cj, cd = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.float32)
    n = fn(N, d.ctypes.data_as(_f32p), C.c_float(thr), mv.ctypes.data_as(_f32p), cap, ci.ctypes.data_as(_i32p),
           cj.ctypes.data_as(_i32p), cd.ctypes.data_as(_f32p))
    return mv, ci[:n].copy(), cj[:n].copy(), cd[:n].copy()


def oracle_scan(d, thr):
    """(min_values, cand_i, cand_j, cand_dist) of the restated Initialize scans."""
    return _scan(_lib().oracle_minmatch_scan, d, thr)


def quick_build(d, theta, sample_ages=None, Ne=30000):
    """(parents[2N-1], number of merges that fell back to the symmetric matrix) of the restated QuickBuild."""
    d = np.ascontiguousarray(d, dtype=np.float32)
    N = d.shape[0]
    par = np.zeros(2 * N - 1, dtype=np.int32)
    a = None if sample_ages is None else np.ascontiguousarray(sample_ages, dtype=np.float64)
    rc = _lib().oracle_quick_build(N, d.ctypes.data_as(_f32p), theta, Ne, None if a is None else a.ctypes.data_as(_f64p),
                                   par.ctypes.data_as(_i32p))
    return par, rc


def initialize(d, theta, sample_ages=None, Ne=30000):
    """mcandidates (lin1, lin2, dist, dist2), best (lin1, lin2, dist, dist2, dist3) and min_values of the restated
    Initialize with the generator seeded as in QuickBuild."""
    d = np.ascontiguousarray(d, dtype=np.float32)
    N = d.shape[0]
    l1, l2 = np.zeros(N, dtype=np.int32), np.zeros(N, dtype=np.int32)
    ds, d2 = np.zeros(N), np.zeros(N)
    bl, bd = np.zeros(2, dtype=np.int32), np.zeros(3)
    mv = np.zeros(N, dtype=np.float32)
    a = None if sample_ages is None else np.ascontiguousarray(sample_ages, dtype=np.float64)
    _lib().oracle_minmatch_initialize(N, d.ctypes.data_as(_f32p), theta, Ne, None if a is None else a.ctypes.data_as(_f64p),
                                      l1.ctypes.data_as(_i32p), l2.ctypes.data_as(_i32p), ds.ctypes.data_as(_f64p),
                                      d2.ctypes.data_as(_f64p), bl.ctypes.data_as(_i32p), bd.ctypes.data_as(_f64p),
                                      mv.ctypes.data_as(_f32p))
    return (l1, l2, ds, d2), (int(bl[0]), int(bl[1]), float(bd[0]), float(bd[1]), float(bd[2])), mv


def uniform_draws(n, seed=1):
    out = np.zeros(n)
    _lib().oracle_uniform_draws(seed, n, out.ctypes.data_as(_f64p))
    return out


_host = None


def hostrun_scan(d, thr):
    """The library's scan kernels (relate_amd/csrc/minmatch_kernels.hpp) run on CPU threads."""
    global _host
    if _host is None:
        so = os.path.join(HERE, "hostrun", "libminmatch_host.so")
        src = [os.path.join(HERE, "hostrun", "minmatch_host.cpp"), os.path.join(HERE, "hostrun", "hip_host_shim.hpp"),
               os.path.join(ROOT, "relate_amd", "csrc", "minmatch_kernels.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
            subprocess.run(["g++", "-std=c++20", "-O1", "-fPIC", "-shared", "-pthread", "-DRELATE_HOST_RUN",
                            "-Wno-unknown-pragmas", "-o", so, src[0]], check=True)
        _host = C.CDLL(so)
        _host.hostrun_minmatch_scan.restype = C.c_longlong
        _host.hostrun_minmatch_scan.argtypes = [C.c_int, _f32p, C.c_float, _f32p, C.c_longlong, _i32p, _i32p, _f32p]
    return _scan(_host.hostrun_minmatch_scan, d, thr)


def distance_like_matrix(N, seed, ties=True):
    """A matrix shaped like GetMatrix's output: non-negative, zero diagonal, every row's minimum zero; with `ties`
    many equal entries (distances are multiples of a mutation's weight), so that candidate pairs are plentiful."""
    rng = np.random.default_rng(seed)
    if ties:
        d = rng.integers(0, 6, (N, N)).astype(np.float32) * np.float32(7.25) + rng.integers(0, 2, (N, N)).astype(np.float32) * np.float32(0.5)
    else:
        d = rng.random((N, N)).astype(np.float32) * 30
    np.fill_diagonal(d, np.inf)
    d -= d.min(axis=1, keepdims=True)
    np.fill_diagonal(d, 0)
    return d.astype(np.float32)
