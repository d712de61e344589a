"""Host-side mirror of the reference's DistanceMeasure (src/anc_builder.hpp:35-90, src/anc_builder.cpp:9-211):
the object AncesTreeBuilder::BuildTopology asks for the distance matrix of a SNP.  The topology of the
current section and the matrix live in HBM (relate_plan_repaint / relate_plan_distance_matrix); what stays on
the host is the reference's own bookkeeping, under its own names: v_snp_prev, v_rpos_prev, v_rpos_next.

    d = DistanceMeasure(plan, seq, rpos, stones, section=0)   # plan: loaded, prepared for all N targets, painted
    m = d.GetMatrix(section_startpos)                          # N x N, on the device
    for snp in range(section_startpos + 1, section_endpos + 1):
        d.advance(snp)                                         # src/anc_builder.cpp:331-335
        ...
        m = d.GetMatrix(snp)

The painter behind the re-painting is the one the plan was prepared with; the reference builds it from Data
(FastPainting painter(*data), src/anc_builder.cpp:59), i.e. prepare the plan with theta_painter == theta_data.
"""
import ctypes as C

import numpy as np

from ._lib import check, f64p, i64p, lib, record_bytes, u8p, vp


class DistanceMeasure:
    def __init__(self, plan, seq, rpos, stones, section=0):
        """seq: Data::sequence (L x N bytes '0'/'1'); rpos: Data::rpos (L+1 doubles); stones: torch uint8 tensor
        on the plan's device holding what plan.paint() wrote (W x N records)."""
        import torch
        self._torch = torch
        self.plan = plan
        self.N, self.L = plan.N, plan.L
        if plan.K != plan.N:
            raise ValueError("the plan must be prepared for all N targets")
        self.derived = np.ascontiguousarray(seq) == ord("1")
        self.rpos = np.asarray(rpos, dtype=np.float64)
        self.stones = stones
        self.section = section
        self.section_startpos = -1
        self.section_endpos = -1
        self.v_snp_prev = np.zeros(self.N, dtype=np.int64)
        self.v_rpos_prev = np.zeros(self.N)
        self.v_rpos_next = np.zeros(self.N)
        self.matrix = torch.zeros((self.N, self.N), dtype=torch.float32, device=stones.device)
        self.topology = self.logscales = None
        self.row_offsets = None
        self.pitch = 0

    def GetTopologyWithRepaint(self, snp):
        """src/anc_builder.cpp:53-110: re-paint the next section for every target, reset the bookkeeping."""
        torch, plan, N = self._torch, self.plan, self.N
        w = self.section
        rows, pitch, off = plan.repaint_layout(w)
        self.topology = torch.empty((rows, pitch), dtype=torch.float32, device=self.stones.device)
        self.logscales = torch.empty((rows,), dtype=torch.float32, device=self.stones.device)
        torch.cuda.synchronize(self.stones.device)
        plan.repaint(w, self.stones.data_ptr() + w * N * record_bytes(N), self.topology.data_ptr(),
                     self.logscales.data_ptr())
        self.row_offsets, self.pitch, self.rows = off, pitch, rows
        # interval of the section as the records carry it (:66-67)
        head = self.stones[w * N * record_bytes(N):w * N * record_bytes(N) + 8].cpu().numpy().view(np.int32)
        self.section_startpos, self.section_endpos = int(head[0]), int(head[1])
        self.v_snp_prev[:] = 0
        if snp > 0:  # :86-97
            self.v_snp_prev[:] = self.derived[self.section_startpos:snp + 1].sum(axis=0)
        for n in range(N):  # :100-105
            t = snp
            while not self.derived[t, n] and t > 0:
                t -= 1
            self.v_rpos_prev[n] = self.rpos[t]
        self.v_rpos_next[:] = self.v_rpos_prev
        assert self.section_startpos <= snp <= self.section_endpos
        self.section += 1

    def advance(self, snp):
        """What BuildTopology does to the bookkeeping when it moves on to `snp` (src/anc_builder.cpp:331-335)."""
        c = self.derived[snp]
        self.v_snp_prev[c] += 1
        self.v_rpos_prev[c] = self.rpos[snp]

    def GetMatrix(self, snp):
        """src/anc_builder.cpp:113-211.  Returns the N x N matrix (torch tensor on the device, also self.matrix)."""
        if snp > self.section_endpos:
            self.GetTopologyWithRepaint(snp)
        N, L = self.N, self.L
        direct = self.derived[snp] | (snp == 0) | (snp == L - 1)
        wl = np.zeros(N)
        wr = np.zeros(N)
        for n in np.nonzero(~direct)[0]:
            if self.v_rpos_next[n] <= self.v_rpos_prev[n]:  # :144-151
                for l in range(snp, L):
                    if self.derived[l, n] or l == L - 1:
                        self.v_rpos_next[n] = self.rpos[l]
                        break
            rp, rn = self.v_rpos_prev[n], self.v_rpos_next[n]
            if rp == rn:
                wl[n] = wr[n] = 0.5
            else:
                wl[n] = (rn - self.rpos[snp]) / (rn - rp)
                wr[n] = (self.rpos[snp] - rp) / (rn - rp)
        self.plan.distance_matrix(N, self.topology.data_ptr(), self.pitch, self.logscales.data_ptr(), self.rows,
                                  self.row_offsets[:N] + self.v_snp_prev, (~direct).astype(np.uint8), wl, wr,
                                  self.matrix.data_ptr())
        self._last_args = (self.row_offsets[:N] + self.v_snp_prev, (~direct).astype(np.uint8), wl, wr)
        return self.matrix


class NativeDistanceMeasure:
    """The same object kept by the library (relate_distance_*, include/relate_gpu.h): what a C++ maintainer links
    against.  Same method names; the matrix is written into a torch tensor of the caller's device."""

    def __init__(self, plan, seq, rpos, stones, section=0):
        import torch
        self._torch = torch
        self.plan, self.stones = plan, stones  # kept alive: the library borrows both
        self.N, self.L = plan.N, plan.L
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        rpos = np.ascontiguousarray(rpos, dtype=np.float64)
        if seq.shape != (self.L, self.N) or rpos.shape != (self.L + 1,):
            raise ValueError("seq must be (L, N) and rpos (L + 1,)")
        self._h = vp()
        check(lib().relate_distance_new(plan._h, seq.ctypes.data_as(u8p), rpos.ctypes.data_as(f64p), vp(stones.data_ptr()),
                                        int(section), C.byref(self._h)))
        self.matrix = torch.zeros((self.N, self.N), dtype=torch.float32, device=stones.device)

    def close(self):
        if getattr(self, "_h", None):
            lib().relate_distance_free(self._h)
            self._h = None

    __del__ = close

    def GetTopologyWithRepaint(self, snp, stream=0):
        check(lib().relate_distance_get_topology_with_repaint(self._h, int(snp), vp(stream)))

    def advance(self, snp):
        check(lib().relate_distance_advance(self._h, int(snp)))

    def GetMatrix(self, snp, stream=0):
        check(lib().relate_distance_get_matrix(self._h, int(snp), vp(self.matrix.data_ptr()), vp(stream)))
        return self.matrix

    def state(self):
        sec, lo, hi = C.c_int(), C.c_int(), C.c_int()
        v = np.zeros(self.N, dtype=np.int64)
        rp, rn = np.zeros(self.N), np.zeros(self.N)
        check(lib().relate_distance_state(self._h, C.byref(sec), C.byref(lo), C.byref(hi), v.ctypes.data_as(i64p),
                                          rp.ctypes.data_as(f64p), rn.ctypes.data_as(f64p)))
        return {"section": sec.value, "section_startpos": lo.value, "section_endpos": hi.value, "v_snp_prev": v,
                "v_rpos_prev": rp, "v_rpos_next": rn}
