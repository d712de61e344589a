"""Host-side mirror of the opening of the reference's tree builder, MinMatch::Initialize
(src/tree_builder.cpp:57-127, with sample ages :130-214): the two scans over the distance matrix run on the device
(relate_plan_minmatch_scan), the candidate bookkeeping that consumes one random draw per pair stays on the host
(relate_minmatch_apply_candidates, with the caller's generator behind a callback)."""
import ctypes as C

import numpy as np

from ._lib import check, f32p, f64p, i32p, lib, vp


class Candidate(C.Structure):
    """struct Candidate (src/tree_builder.hpp:12-25) = relate_candidate."""
    _fields_ = [("lin1", C.c_int32), ("lin2", C.c_int32), ("dist", C.c_double), ("dist2", C.c_double),
                ("dist3", C.c_double), ("replace", C.c_int32)]


def minmatch_threshold(theta):
    """MinMatch::threshold for Data::theta (src/tree_builder.cpp:43), a float."""
    return float(lib().relate_minmatch_threshold(float(theta)))


class MinMatchScan:
    """min_values and the ordered candidate pairs of an N x N distance matrix that lies on the device."""

    def __init__(self, plan, N, d_matrix, threshold, stream=0):
        self.N = N
        self.min_values = np.zeros(N, dtype=np.float32)
        cap = max(4 * N, 1024)
        while True:
            ci = np.zeros(cap, dtype=np.int32)
            cj = np.zeros(cap, dtype=np.int32)
            cd = np.zeros(cap, dtype=np.float32)
            n = C.c_int64()
            rc = lib().relate_plan_minmatch_scan(plan._h, N, vp(d_matrix), float(threshold),
                                                 self.min_values.ctypes.data_as(f32p), cap, ci.ctypes.data_as(i32p),
                                                 cj.ctypes.data_as(i32p), cd.ctypes.data_as(f32p), C.byref(n), vp(stream))
            if rc == -5 and n.value > cap:  # RELATE_ERANGE: more pairs than the arrays hold
                cap = n.value
                continue
            check(rc)
            break
        self.count = n.value
        self.cand_i, self.cand_j, self.cand_dist = ci[:n.value], cj[:n.value], cd[:n.value]


def apply_candidates(N, cand_i, cand_j, cand_dist, draw, sample_ages=None, age=0.0):
    """The host half of Initialize: (mcandidates, best_candidate) after every pair of the list has been offered,
    `draw()` standing for the caller's dist_unif(rng)."""
    cand_i = np.ascontiguousarray(cand_i, dtype=np.int32)
    cand_j = np.ascontiguousarray(cand_j, dtype=np.int32)
    cand_dist = np.ascontiguousarray(cand_dist, dtype=np.float32)
    ages = None if sample_ages is None else np.ascontiguousarray(sample_ages, dtype=np.float64)
    m = (Candidate * N)()
    inf = float("inf")
    for c in m:
        c.lin1 = c.lin2 = -1
        c.dist = c.dist2 = c.dist3 = inf
    best = Candidate(-1, -1, inf, inf, inf, 0)
    cb = C.CFUNCTYPE(C.c_double, C.c_void_p)(lambda _ctx: float(draw()))
    check(lib().relate_minmatch_apply_candidates(N, len(cand_i), cand_i.ctypes.data_as(i32p), cand_j.ctypes.data_as(i32p),
                                                 cand_dist.ctypes.data_as(f32p),
                                                 None if ages is None else ages.ctypes.data_as(f64p), float(age),
                                                 C.cast(cb, C.c_void_p), None, C.cast(m, C.c_void_p), C.cast(C.pointer(best), C.c_void_p)))
    return m, best
