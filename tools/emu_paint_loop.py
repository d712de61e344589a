#!/usr/bin/env python3
"""CPU-side check of the generated assembly loop of the painting kernel (tools/gen_paint_loop.py).

The loop is one hand-written gfx950 instruction stream per geometry; a wrong register number or a fetch one
element past a list is a GPU memory fault, not a test failure.  This module interprets that very text (the list
of instruction strings the generator emits) on the CPU, wave by wave:

  * every memory access (scalar loads of row words and site records, the vector touch of a row, LDS reads and
    writes) is checked against the extent of the buffer it must fall into -- the buffers are laid out as the
    library allocates them, with nothing mapped in between;
  * scalar loads are asynchronous: a register that a load has not delivered yet (no s_waitcnt lgkmcnt(0) since)
    may be neither read nor overwritten;
  * the arithmetic is IEEE double, as on the device, so the stones the emulated kernel produces can be compared
    with the reference's records bit for bit.

What is NOT taken from the generated text is the C++ around the loop (paint_loop.hpp): the initial state, the
fixed registers it sets, and the rare paths (rescale, stones) are restated here in Python, following that file
statement by statement.

    python3 tools/emu_paint_loop.py [golden ...] [--waves G] [--targets a:b]
"""
import argparse
import math
import os
import re
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gen_paint_loop as gen  # noqa: E402

U32 = 0xFFFFFFFF
U64 = 0xFFFFFFFFFFFFFFFF
EXEC_LO, EXEC_HI, VCC_LO, VCC_HI, M0 = 126, 127, 106, 107, 124  # slots of the special registers in the file below


class EmuError(Exception):
    pass


class Memory:
    """Named buffers at addresses far apart; an access must lie inside one of them."""

    def __init__(self):
        self.bufs = []  # (name, base, bytes-like, readable_len)
        self.next = 0x7F00_0000_0000

    def add(self, name, data, alloc_bytes=None):
        data = bytes(data)
        n = len(data) if alloc_bytes is None else alloc_bytes
        if n < len(data):
            raise ValueError("allocation smaller than its contents")
        base = self.next
        self.next += ((n + 0xFFFF) // 0x10000 + 16) * 0x10000  # unmapped pages between allocations
        self.bufs.append((name, base, data + bytes(n - len(data)), n))
        return base

    def read(self, addr, nbytes, what):
        for name, base, data, n in self.bufs:
            if base <= addr and addr + nbytes <= base + n:
                return data[addr - base:addr - base + nbytes]
        near = min(self.bufs, key=lambda b: min(abs(addr - b[1]), abs(addr - (b[1] + b[3]))))
        raise EmuError(f"{what}: {nbytes} bytes at {addr:#x} are outside every buffer "
                       f"(nearest: {near[0]} [{near[1]:#x}, {near[1] + near[3]:#x}), offset {addr - near[1]})")


class Wave:
    """Register state of one wave and an interpreter for the instruction subset the generator uses."""

    def __init__(self, mem, lds, lds_bytes, name=""):
        self.s = np.zeros(128, dtype=np.uint64)  # 32-bit values; uint64 to keep arithmetic simple
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.scc = 0
        self.mem, self.lds, self.lds_bytes = mem, lds, lds_bytes
        self.pending_s = {}   # sgpr -> value, delivered at s_waitcnt lgkmcnt(0)
        self.pending_v = {}   # (vgpr) -> array, delivered at s_waitcnt (LDS reads) / vmcnt
        self.gpr_idx = None
        self.name = name
        self.stats = {"valu": 0, "salu": 0, "smem": 0, "lds": 0, "vmem": 0}
        self.set_s64(EXEC_LO, U64)

    # ---- register access
    def get_s32(self, r):
        if r in self.pending_s:
            raise EmuError(f"{self.name}: s{r} is read before the scalar load that fills it has been waited for")
        return int(self.s[r]) & U32

    def set_s32(self, r, val):
        if r in self.pending_s:
            raise EmuError(f"{self.name}: s{r} is overwritten while a scalar load into it is in flight")
        self.s[r] = val & U32

    def get_s64(self, r):
        return self.get_s32(r) | (self.get_s32(r + 1) << 32)

    def set_s64(self, r, val):
        self.set_s32(r, val & U32)
        self.set_s32(r + 1, (val >> 32) & U32)

    def exec_mask(self):
        m = self.get_s64(EXEC_LO)
        return np.array([(m >> i) & 1 for i in range(64)], dtype=bool)

    def get_v64(self, r):
        return (self.v[r].astype(np.uint64) | (self.v[r + 1].astype(np.uint64) << np.uint64(32))).view(np.float64)

    def set_v64(self, r, val, mask=None):
        bits = np.asarray(val, dtype=np.float64).view(np.uint64)
        lo, hi = (bits & np.uint64(U32)).astype(np.uint32), (bits >> np.uint64(32)).astype(np.uint32)
        if mask is None:
            mask = self.exec_mask()
        self.v[r][mask] = lo[mask]
        self.v[r + 1][mask] = hi[mask]

    # ---- operands
    _sreg = re.compile(r"^s(\d+)$|^s\[(\d+):(\d+)\]$")
    _vreg = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")

    def sreg(self, tok):
        tok = tok.strip()
        m = self._sreg.match(tok)
        if m:
            return int(m.group(1) if m.group(1) is not None else m.group(2))
        return {"exec": EXEC_LO, "exec_lo": EXEC_LO, "exec_hi": EXEC_HI, "vcc": VCC_LO, "vcc_lo": VCC_LO, "vcc_hi": VCC_HI}.get(tok)

    def vreg(self, tok):
        m = self._vreg.match(tok.strip())
        if not m:
            return None
        return int(m.group(1) if m.group(1) is not None else m.group(2))

    def src32(self, tok):
        tok = tok.strip()
        r = self.sreg(tok)
        if r is not None:
            return self.get_s32(r)
        return int(tok, 0) & U32

    def src64(self, tok):
        tok = tok.strip()
        r = self.sreg(tok)
        if r is not None:
            return self.get_s64(r)
        return int(tok, 0) & U64  # inline constants sign-extend: -1 is all ones

    def vsrc_f64(self, tok):
        """a 64-bit vector source: VGPR pair, SGPR pair (broadcast) or a constant"""
        tok = tok.strip()
        r = self.vreg(tok)
        if r is not None:
            return self.get_v64(r)
        r = self.sreg(tok)
        if r is not None:
            return np.full(64, struct.unpack("<d", struct.pack("<Q", self.get_s64(r)))[0])
        return np.full(64, float(tok))

    # ---- execution: returns None, or ("branch", label), ("barrier",)
    def step(self, line):
        line = line.strip()
        if not line or line.endswith(":"):
            return None
        op, _, rest = line.partition(" ")
        rest = rest.strip()
        if op.startswith("v_mov_b32_dpp"):
            return self.op_dpp(rest)
        a = [x.strip() for x in rest.split(",")] if rest else []
        f = getattr(self, "op_" + op, None)
        if f is None:
            raise EmuError(f"instruction not modelled: {line}")
        return f(*a)

    # scalar ALU
    def _salu(self):
        self.stats["salu"] += 1

    def op_s_mov_b32(self, d, a):
        self._salu()
        self.set_s32(self.sreg(d), self.src32(a))

    def op_s_mov_b64(self, d, a):
        self._salu()
        self.set_s64(self.sreg(d), self.src64(a))

    def op_s_and_b32(self, d, a, b):
        self._salu()
        r = self.src32(a) & self.src32(b)
        self.set_s32(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_lshr_b32(self, d, a, b):
        self._salu()
        r = self.src32(a) >> (self.src32(b) & 31)
        self.set_s32(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_add_u32(self, d, a, b):
        self._salu()
        r = self.src32(a) + self.src32(b)
        self.set_s32(self.sreg(d), r)
        self.scc = int(r > U32)

    def op_s_addc_u32(self, d, a, b):
        self._salu()
        r = self.src32(a) + self.src32(b) + self.scc
        self.set_s32(self.sreg(d), r)
        self.scc = int(r > U32)

    def op_s_sub_u32(self, d, a, b):
        self._salu()
        x, y = self.src32(a), self.src32(b)
        self.set_s32(self.sreg(d), x - y)
        self.scc = int(y > x)

    def op_s_mul_i32(self, d, a, b):
        self._salu()
        self.set_s32(self.sreg(d), self.src32(a) * self.src32(b))

    def op_s_mul_hi_u32(self, d, a, b):
        self._salu()
        self.set_s32(self.sreg(d), (self.src32(a) * self.src32(b)) >> 32)

    def op_s_andn2_b64(self, d, a, b):
        self._salu()
        r = self.src64(a) & ~self.src64(b) & U64
        self.set_s64(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_cselect_b32(self, d, a, b):
        self._salu()
        self.set_s32(self.sreg(d), self.src32(a) if self.scc else self.src32(b))

    def op_s_cselect_b64(self, d, a, b):
        self._salu()
        self.set_s64(self.sreg(d), self.src64(a) if self.scc else self.src64(b))

    def _cmp(self, a, b, f):
        self._salu()
        self.scc = int(f(self.src32(a), self.src32(b)))

    def op_s_cmp_eq_u32(self, a, b):
        self._cmp(a, b, lambda x, y: x == y)

    def op_s_cmp_lg_u32(self, a, b):
        self._cmp(a, b, lambda x, y: x != y)

    def op_s_cmp_ge_u32(self, a, b):
        self._cmp(a, b, lambda x, y: x >= y)

    def op_s_cmp_lt_u32(self, a, b):
        self._cmp(a, b, lambda x, y: x < y)

    def op_s_cbranch_scc0(self, label):
        return ("branch", label) if not self.scc else None

    def op_s_cbranch_scc1(self, label):
        return ("branch", label) if self.scc else None

    def op_s_branch(self, label):
        return ("branch", label)

    def op_s_setprio(self, a):
        pass

    def op_s_nop(self, a):
        pass

    def op_s_barrier(self):
        return ("barrier",)

    def op_s_waitcnt(self, *what):
        text = " ".join(what)
        if "lgkmcnt(0)" in text:
            for r, val in self.pending_s.items():
                self.s[r] = val
            self.pending_s.clear()
            for r, val in self.pending_v.items():
                self.v[r] = val
            self.pending_v.clear()

    def op_s_set_gpr_idx_on(self, a, mode):
        self._salu()
        if int(mode, 0) != 8:
            raise EmuError("only destination-relative index mode is modelled")
        self.gpr_idx = self.src32(a) & 0xFF

    def op_s_set_gpr_idx_off(self):
        self._salu()
        self.gpr_idx = None

    # scalar memory
    def _sload(self, d, base, off, ndw):
        self.stats["smem"] += 1
        r0 = self.sreg(d)
        addr = self.src64(base) + int(off, 0)
        if addr % 4:
            raise EmuError(f"{self.name}: unaligned scalar load")
        data = self.mem.read(addr, 4 * ndw, f"{self.name}: s_load_dwordx{ndw} into s[{r0}:{r0 + ndw - 1}]")
        for i in range(ndw):
            if r0 + i in self.pending_s:
                raise EmuError(f"{self.name}: two scalar loads in flight into s{r0 + i}")
            self.pending_s[r0 + i] = struct.unpack_from("<I", data, 4 * i)[0]

    def op_s_load_dwordx2(self, d, base, off):
        self._sload(d, base, off, 2)

    def op_s_load_dwordx4(self, d, base, off):
        self._sload(d, base, off, 4)

    def op_s_load_dwordx8(self, d, base, off):
        self._sload(d, base, off, 8)

    def op_s_load_dwordx16(self, d, base, off):
        self._sload(d, base, off, 16)

    # vector ALU
    def op_v_add_f64(self, d, a, b):
        self.stats["valu"] += 1
        self.set_v64(self.vreg(d), self.vsrc_f64(a) + self.vsrc_f64(b))

    def op_v_mul_f64(self, d, a, b):
        self.stats["valu"] += 1
        self.set_v64(self.vreg(d), self.vsrc_f64(a) * self.vsrc_f64(b))

    def op_v_mov_b64(self, d, a):
        self.stats["valu"] += 1
        self.set_v64(self.vreg(d), self.vsrc_f64(a))

    def _v32(self, tok):
        r = self.vreg(tok)
        if r is not None:
            return self.v[r].copy()
        return np.full(64, self.src32(tok), dtype=np.uint32)

    def op_v_xor_b32(self, d, a, b):
        self.stats["valu"] += 1
        m = self.exec_mask()
        self.v[self.vreg(d)][m] = (self._v32(a) ^ self._v32(b))[m]

    def op_v_mov_b32(self, d, a):
        self.stats["valu"] += 1
        dst = self.vreg(d) + (self.gpr_idx or 0)
        if dst > 255:
            raise EmuError(f"{self.name}: indexed VGPR write past v255")
        m = self.exec_mask()
        self.v[dst][m] = self._v32(a)[m]

    def op_v_readlane_b32(self, d, a, lane):
        self.stats["valu"] += 1
        self.set_s32(self.sreg(d), int(self.v[self.vreg(a)][int(lane, 0)]))

    def op_v_readfirstlane_b32(self, d, a):
        self.stats["valu"] += 1
        m = self.exec_mask()
        lane = int(np.argmax(m)) if m.any() else 0
        self.set_s32(self.sreg(d), int(self.v[self.vreg(a)][lane]))

    def op_dpp(self, rest):
        self.stats["valu"] += 1
        m = re.match(r"v(\d+),\s*v(\d+)\s+(.*?)\s+row_mask:(\w+)\s+bank_mask:(\w+)$", rest)
        if not m:
            raise EmuError("DPP form not modelled: " + rest)
        d, a, ctrl, row_mask, bank_mask = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4), 0), int(m.group(5), 0)
        src, out = self.v[a].copy(), self.v[d].copy()
        ex = self.exec_mask()
        for lane in range(64):
            row, inrow = lane >> 4, lane & 15
            if not ex[lane] or not (row_mask >> row) & 1 or not (bank_mask >> (inrow >> 2)) & 1:
                continue
            if ctrl.startswith("quad_perm:"):
                p = [int(x) for x in ctrl[len("quad_perm:["):-1].split(",")]
                sl = (lane & ~3) | p[lane & 3]
            elif ctrl == "row_half_mirror":
                sl = (lane & ~7) | (7 - (lane & 7))
            elif ctrl == "row_mirror":
                sl = (lane & ~15) | (15 - inrow)
            elif ctrl == "row_bcast:15":
                if row == 0:
                    continue  # no source: the lane keeps its value
                sl = 16 * (row - 1) + 15
            elif ctrl == "row_bcast:31":
                if row < 2:
                    continue
                sl = 31
            else:
                raise EmuError("DPP control not modelled: " + ctrl)
            if not ex[sl]:
                continue  # a disabled source lane leaves the destination alone (bound_ctrl is off)
            out[lane] = src[sl]
        self.v[d] = out

    # vector memory and LDS
    def op_global_load_dword(self, d, voff, sbase, *mods):
        self.stats["vmem"] += 1
        off = 0
        for mod in mods + tuple(sbase.split()[1:]):
            if mod.startswith("offset:"):
                off = int(mod[7:], 0)
        base = self.src64(sbase.split()[0])
        ex = self.exec_mask()
        for lane in range(64):
            if ex[lane]:
                self.mem.read(base + int(self.v[self.vreg(voff)][lane]) + off, 4, f"{self.name}: global_load_dword lane {lane}")
        # the value is never used (a cache touch): leave the register alone

    def _lds_addr(self, tok, lane, off, nbytes):
        a = int(self.v[self.vreg(tok)][lane]) + off
        if a < 0 or a + nbytes > self.lds_bytes or a % 8:
            raise EmuError(f"{self.name}: LDS access of {nbytes} bytes at {a} (allocation {self.lds_bytes})")
        return a

    def op_ds_write_b64(self, addr, data):
        self.stats["lds"] += 1
        ex = self.exec_mask()
        r = self.vreg(data)
        for lane in range(64):
            if ex[lane]:
                a = self._lds_addr(addr, lane, 0, 8)
                struct.pack_into("<II", self.lds, a, int(self.v[r][lane]), int(self.v[r + 1][lane]))

    def op_ds_read_b128(self, d, addr_and_mods):
        self.stats["lds"] += 1
        parts = addr_and_mods.split()
        off = 0
        for mod in parts[1:]:
            if mod.startswith("offset:"):
                off = int(mod[7:], 0)
        r = self.vreg(d)
        ex = self.exec_mask()
        vals = [self.v[r + i].copy() for i in range(4)]
        for lane in range(64):
            if ex[lane]:
                a = self._lds_addr(parts[0], lane, off, 16)
                w = struct.unpack_from("<IIII", self.lds, a)
                for i in range(4):
                    vals[i][lane] = w[i]
        for i in range(4):
            self.pending_v[r + i] = vals[i]


def run_block(waves, lines, labels, start, inputs):
    """Runs the instruction list on all waves of a workgroup from `start` (a label or 0) to the end of the list;
    the waves stop at every s_barrier until all have arrived.  `inputs`: per wave, the values of the asm operands."""
    pcs = [labels[start] if isinstance(start, str) else start for _ in waves]
    done = [False] * len(waves)
    budget = 50_000_000
    while not all(done):
        at_barrier = []
        for wi, w in enumerate(waves):
            if done[wi]:
                continue
            while True:
                budget -= 1
                if budget < 0:
                    raise EmuError("instruction budget exhausted: the loop does not terminate")
                if pcs[wi] >= len(lines):
                    done[wi] = True
                    break
                line = lines[pcs[wi]]
                for key, val in inputs[wi].items():
                    line = line.replace(key, val)
                pcs[wi] += 1
                res = w.step(line)
                if res is None:
                    continue
                if res[0] == "branch":
                    pcs[wi] = labels[res[1]]
                elif res[0] == "barrier":
                    at_barrier.append(wi)
                    break
        if at_barrier and len(at_barrier) != sum(1 for d in done if not d):
            raise EmuError("some waves of the workgroup finished while others wait at a barrier")


def index_labels(lines):
    return {ln.strip()[:-1]: i for i, ln in enumerate(lines) if ln.strip().endswith(":")}


# ------------------------------------------------------------------------------------------------------------
# The kernel around the loop (paint_loop.hpp), restated.
# ------------------------------------------------------------------------------------------------------------
def fast_log(x):
    """src/fast_log.hpp:6-21 on a float32 (numpy scalar arithmetic rounds every operation to float)."""
    f = np.float32
    bits = int(np.array(x, dtype=np.float32).view(np.uint32))
    e = ((bits >> 23) & 255) - 128
    bits = (bits & ~(255 << 23) & U32) + (127 << 23)
    m = np.array(bits, dtype=np.uint32).view(np.float32)
    t = f(f(-1.0) / f(3)) * m
    t = t + f(2.0)
    t = t * m
    t = t - f(f(2.0) / f(3))
    t = t + f(e)
    return float(t * f(0.69314718))


def row_stride_words(N):
    return (((N + 63) // 64) + 7) // 8 * 8  # panels of up to 80 words (wider ones pad further; not emulated here)


class Chunk:
    """Inputs of one chunk as the library lays them out in HBM (paint_kernels.hip, relate_gpu.cpp)."""

    def __init__(self, seq, r, wb, theta, theta_d=None):
        theta_d = theta if theta_d is None else theta_d
        self.L, self.N = seq.shape
        L, N = self.L, self.N
        self.W = len(wb) - 1
        self.wb = np.asarray(wb, dtype=np.int64)
        self.theta = theta
        self.stride = row_stride_words(N)
        bits = np.ones((L, self.stride * 64), dtype=np.uint8)  # bits from N on are set (pack_rows)
        bits[:, :N] = seq == ord("1")
        self.rows = np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(L, self.stride)
        self.der = seq == ord("1")
        self.r = np.asarray(r, dtype=np.float64)
        Nm1 = float(N - 1)
        ntheta = 1.0 - theta
        self.c = dict(Nm1=Nm1, prior_mis=1.0 * (theta / Nm1 - ntheta / Nm1) + ntheta / Nm1, prior_match=ntheta / Nm1,
                      rho=1.0 * (theta / ntheta - 1.0) + 1.0, log_ntheta=math.log(ntheta), log_small=math.log(0.01),
                      log_Nm1=math.log(Nm1), theta_d=theta_d, ntheta_d=1.0 - theta_d, inv_theta_d=1.0 / theta_d,
                      inv_ntheta_d=1.0 / (1.0 - theta_d), lo=1e-10, hi=1.0 / 1e-10, same_theta=int(theta == theta_d))
        # per target lists
        self.dk, self.rec, self.off = [], [], []
        self.bbj = np.zeros((N, self.W), dtype=np.int64)
        self.bej = np.zeros((N, self.W), dtype=np.int64)
        rec_bytes = bytearray()
        o = 0
        for k in range(N):
            d = np.nonzero(self.der[:, k])[0]
            d = np.unique(np.concatenate([[0], d, [L - 1]]))
            nd = len(d)
            recs = []
            for j in range(nd):
                s0 = int(d[j])
                s1 = int(d[j + 1]) if j + 1 < nd else s0 + 1
                rsum = float(self.r[s0])
                for s in range(s0 + 1, s1):
                    rsum += float(self.r[s])
                nxv = -rsum + self.c["log_ntheta"]
                rp = 1.0 - math.exp(-rsum)
                if rp > 0.99:
                    rp = 0.99
                    nxv = self.c["log_small"] + self.c["log_ntheta"]
                cf = rp / ((1.0 - rp) * Nm1)
                f2 = int(d[min(j + 2, nd - 1)])
                b2 = int(d[max(j - 2, 0)])
                recs.append((cf, nxv, f2, b2))
                rec_bytes += struct.pack("<ddii", cf, nxv, f2, b2)
            self.dk.append(d)
            self.rec.append(recs)
            self.off.append(o)
            o += nd
            self.bbj[k, 0] = 0
            self.bej[k, self.W - 1] = nd - 1
            for w in range(1, self.W):
                lo = int(np.searchsorted(d, self.wb[w], side="left"))
                self.bej[k, w - 1] = lo
                self.bbj[k, w] = lo - 1
        self.mem = Memory()
        rows_alloc = (L * self.stride + 8 * 80 + 64) * 8
        # DevBuf::reserve allocates bytes + bytes / 8 + 256
        self.rows_base = self.mem.add("rows", self.rows.tobytes(), rows_alloc + rows_alloc // 8 + 256)
        self.rec_base = self.mem.add("rec", rec_bytes, len(rec_bytes) + len(rec_bytes) // 8 + 256)


def f64_bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def paint_target_pass(ch, k, pass_, MPL, G, body, labels, collect_stats=None):
    """One workgroup of paint_kernel_a<MPL, G>: target k, pass 0 (forward) or 1 (backward on g = beta * e).
    Returns the stones {window: (site, logscale float32, float32[N])}."""
    S, V, A0 = gen.S, gen.V, gen.A0
    N, L, W, stride, c = ch.N, ch.L, ch.W, ch.stride, ch.c
    dk, rec, nd = ch.dk[k], ch.rec[k], len(ch.dk[k])
    rec_ptr = ch.rec_base + 24 * ch.off[k]
    kw, kb = k >> 6, k & 63
    lds_bytes = G * MPL * 256 + 3 * G * 8
    lds = bytearray(lds_bytes)
    waves = [Wave(ch.mem, lds, lds_bytes, f"target {k} pass {pass_} wave {g}") for g in range(G)]
    lanes = np.arange(64)
    sk_first = U64 if (int(ch.rows[0, kw]) >> kb) & 1 else 0
    sk_last = U64 if (int(ch.rows[L - 1, kw]) >> kb) & 1 else 0

    def row_of(wv, site):
        return ch.rows_base + 8 * (wv * MPL) + site * stride * 8

    # ---- initial states and their sum (the C++ sums with block_sum: a tree; any order gives these bits or an ulp
    # away, which the comparison with the oracle tolerates exactly as the GPU tests do)
    init_row = 0 if pass_ == 0 else L - 1
    sk0 = sk_first if pass_ == 0 else sk_last
    total = 0.0
    for g, w in enumerate(waves):
        for i in range(MPL):
            wi = g * MPL + i
            n = wi * 64 + lanes
            valid = (n < N) & (n != k)
            word = int(ch.rows[init_row, wi]) if wi < stride else U64
            mis = np.array([((~word & sk0) >> l) & 1 for l in range(64)], dtype=bool)
            if pass_ == 0:
                a = np.where(mis, c["prior_mis"], c["prior_match"])
            else:
                a = np.where(mis, c["theta_d"], c["ntheta_d"])
            a = np.where(valid, a, 0.0)
            w.set_v64(A0 + 2 * i, a)
            total += float(a.sum())
    stones = {}
    for g, w in enumerate(waves):
        full_words, rem = N >> 6, N & 63
        wbase = g * MPL
        self_here = wbase <= kw < wbase + MPL
        w_ph = max(full_words, wbase)
        pad_here = w_ph - wbase < MPL
        w.set_v64(V["S"], np.full(64, total))
        w.set_v64(V["LS"], np.full(64, 0.0 if pass_ == 0 else c["log_Nm1"] - float(nd) * c["log_ntheta"]))
        w.set_s64(S["SELFM"], (1 << kb) if self_here else 0)
        w.set_s32(S["SELFIDX"], 2 * (kw - wbase) if self_here else 0)
        w.set_s64(S["PHM"], (~((1 << rem) - 1) & U64) if (w_ph == full_words and rem) else U64)
        w.set_s32(S["PHIDX"], 2 * (w_ph - wbase) if pad_here else 2 * MPL)
        w.set_s32(S["HASSL"], 1 if (self_here or pad_here) else 0)
        w.set_s64(S["ROWSW"], row_of(g, 0))
        w.set_s32(S["STRIDE"], 8 * stride)
        w.set_s32(S["RANK"], 0)
        w.set_v64(V["RHO"], np.full(64, c["rho"]))
        w.v[V["LANEB"]] = (4 * lanes).astype(np.uint32)
        if G > 1:
            w.v[V["LDSR"]] = np.uint32(8 * G)
            w.v[V["LDSW"]] = np.uint32(8 * G + 8 * g)

    def states(w):
        return np.stack([w.get_v64(A0 + 2 * i) for i in range(MPL)])

    def all_states():
        return np.concatenate([states(w).reshape(-1) for w in waves])[:N]

    def ls_now():
        return float(waves[0].get_v64(V["LS"])[0])

    rr = gen.rounds(MPL)

    def asm(w, lines):
        for ln in lines:
            w.step(ln)

    if pass_ == 0:
        wcur = 0
        stone_left = nd - 1

        def emit(left):
            nonlocal wcur, stone_left
            while stone_left == left:
                stones[wcur] = (int(dk[nd - 1 - left]), np.float32(ls_now()), all_states().astype(np.float32))
                wcur += 1
                stone_left = nd - 1 - int(ch.bbj[k, wcur]) if wcur < W else -1
        emit(nd - 1)
        for g, w in enumerate(waves):
            w.set_s32(S["DIR"], 0)
            w.set_s64(S["RECSTEP"], 24)
            w.set_s64(S["SKEDGE"], sk_last)
            w.set_s64(S["COEF"], f64_bits(rec[0][0]))
            w.set_s64(S["NX"], f64_bits(rec[0][1]))
            w.set_s64(S["RP"], row_of(g, int(dk[1])))
            w.set_s64(S["RPNEXT"], row_of(g, int(dk[2 if 2 < nd else nd - 1])))
            w.set_s64(S["TMP"], w.get_s64(S["RPNEXT"]))
            w.set_s64(S["REC"], rec_ptr + 24 * 1)
            w.set_s32(S["LEFT"], nd - 2)
            w.set_s32(S["STONELEFT"], stone_left & U32)
            asm(w, gen.sload("RP", rr[0][0], rr[0][1]))
            asm(w, gen.fetch_record())
    else:
        wcur = W - 1
        stone_j = nd - 1
        while stone_j == nd - 1:
            stones[wcur] = (L - 1, np.float32(ls_now()), np.ones(N, dtype=np.float32))
            wcur -= 1
            stone_j = int(ch.bej[k, wcur]) if wcur >= 0 else -1
        for g, w in enumerate(waves):
            w.set_s32(S["DIR"], 1)
            w.set_s64(S["RECSTEP"], (-24) & U64)
            w.set_s64(S["SKEDGE"], sk_first)
            w.set_s64(S["COEF"], f64_bits(rec[nd - 1][0]))
            w.set_s64(S["NX"], f64_bits(rec[nd - 1][1]))
            w.set_s64(S["RP"], row_of(g, int(dk[nd - 2])))
            w.set_s64(S["RPNEXT"], row_of(g, int(dk[nd - 3 if nd > 2 else 0])))
            w.set_s64(S["TMP"], w.get_s64(S["RPNEXT"]))
            w.set_s64(S["REC"], rec_ptr + 24 * (nd - 2))
            w.set_s32(S["LEFT"], nd - 2)
            w.set_s32(S["STONELEFT"], stone_j & U32)
            asm(w, gen.sload("RP", rr[0][0], rr[0][1]))
            asm(w, gen.fetch_record())

    resume = 0
    sites = 0
    while True:
        # output operands of the asm statement: two SGPRs of the compiler's; any two free slots of the file do
        inputs = [{"%[resume]": str(resume), "%[left]": "s110", "%[shi]": "s111", "%=": ""} for _ in waves]
        run_block(waves, body, labels, 0, inputs)
        lefts = {int(w.get_s32(110)) for w in waves}
        if len(lefts) != 1:
            raise EmuError("the waves of a workgroup left the loop at different sites")
        left = lefts.pop()
        left = left - (1 << 32) if left >> 31 else left
        if left < 0:
            break
        resume = 1
        shi = int(waves[0].get_s32(111))
        if not ((shi - 0x3DE00000) & U32) < (0x42000000 - 0x3DE00000):
            Ssum = float(waves[0].get_v64(V["S"])[0])
            if Ssum < c["lo"] or Ssum > c["hi"]:
                inv = 1.0 / Ssum
                for w in waves:
                    for i in range(MPL):
                        w.set_v64(A0 + 2 * i, w.get_v64(A0 + 2 * i) * inv)
                    w.set_v64(V["S"], np.full(64, 1.0))
                    add = math.log(Ssum) if pass_ == 0 else fast_log(np.float32(Ssum))
                    w.set_v64(V["LS"], w.get_v64(V["LS"]) + add)
        if pass_ == 0:
            emit(left)
            for w in waves:
                w.set_s32(S["STONELEFT"], stone_left & U32)
        else:
            j = left
            while stone_j == j:
                vals = []
                for g, w in enumerate(waves):
                    w.step("s_waitcnt lgkmcnt(0)")
                    skm = w.get_s64(S["SKM"])
                    st = states(w)
                    for (w0, nw) in rr:
                        asm(w, gen.sload("RP", w0, nw))
                        w.step("s_waitcnt lgkmcnt(0)")
                        for i in range(nw):
                            word = w.get_s64(S["BUF"] + 2 * i)
                            mis = np.array([((skm & ~word) >> l) & 1 for l in range(64)], dtype=bool)
                            vals.append(np.where(mis, st[w0 + i] * c["inv_theta_d"], st[w0 + i] * c["inv_ntheta_d"]))
                    w.step("s_waitcnt lgkmcnt(0)")
                    asm(w, gen.sload("RPNEXT", rr[0][0], rr[0][1]))
                stones[wcur] = (int(dk[j]), np.float32(ls_now()), np.concatenate(vals)[:N].astype(np.float32))
                wcur -= 1
                stone_j = int(ch.bej[k, wcur]) if wcur >= 0 else -1
            for w in waves:
                w.set_s32(S["STONELEFT"], stone_j & U32)
        sites += 1
    for w in waves:
        w.step("s_waitcnt lgkmcnt(0)")
    if collect_stats is not None:
        for w in waves:
            for key, val in w.stats.items():
                collect_stats[key] = collect_stats.get(key, 0) + val
        collect_stats["wave_sites"] = collect_stats.get("wave_sites", 0) + G * (nd - 1)
    return stones


def check_case(seq, r, wb, theta, want, MPL, G, targets, verbose=True):
    """Emulates the kernel for `targets` and compares every stone with the records `want` (W, N, 56 + 8N bytes)."""
    import oracle_lib as ol
    ch = Chunk(seq, r, wb, theta)
    if G * MPL * 64 < ch.N or G * MPL > ch.stride:
        raise ValueError("geometry does not fit the panel")
    body = gen.loop_body(MPL, G)
    labels = index_labels([ln.replace("%=", "") for ln in body])
    d = ol.decode_records(want)
    stats = {}
    nbad = nfloat = 0
    for k in targets:
        for pass_ in (0, 1):
            stones = paint_target_pass(ch, k, pass_, MPL, G, body, labels, stats)
            if sorted(stones) != list(range(ch.W)):
                raise EmuError(f"target {k} pass {pass_}: stones of windows {sorted(stones)}")
            for w, (site, ls, vals) in stones.items():
                snp, wls, wv = ((d["a_snp"], d["a_ls"], d["alpha"]) if pass_ == 0 else (d["b_snp"], d["b_ls"], d["beta"]))
                if site != int(snp[w, k]):
                    raise EmuError(f"target {k} pass {pass_} window {w}: stone at site {site}, reference {int(snp[w, k])}")
                nfloat += vals.size + 1
                nbad += int(np.count_nonzero(vals.view(np.uint32) != wv[w, k].view(np.uint32)))
                nbad += int(np.float32(ls).view(np.uint32) != wls[w, k].view(np.uint32))
                if not np.allclose(vals, wv[w, k], rtol=1e-5, atol=0) or abs(float(ls) - float(wls[w, k])) > 1e-5 * max(1.0, abs(float(wls[w, k]))):
                    raise EmuError(f"target {k} pass {pass_} window {w}: values differ from the reference beyond 1e-5")
    if verbose:
        ws = max(stats.get("wave_sites", 1), 1)
        print(f"  {len(targets)} targets x 2 passes as {G} x {MPL} words: every address inside its buffer, no register "
              f"used before its load was waited for; {nbad} of {nfloat} floats not bit-identical to the reference; "
              f"per wave and site: {stats['valu'] / ws:.1f} vector, {stats['salu'] / ws:.1f} scalar, "
              f"{stats['smem'] / ws:.1f} scalar-memory instructions")
    return nbad, nfloat, stats


def main():
    from conftest import load_golden
    ap = argparse.ArgumentParser()
    ap.add_argument("golden", nargs="*", default=["n130_tail"])
    ap.add_argument("--waves", type=int, default=1)
    ap.add_argument("--mpl", type=int, default=0)
    ap.add_argument("--targets", default="")
    args = ap.parse_args()
    for name in args.golden:
        g = load_golden(name)
        if g["theta_painter"] != g["theta_data"]:
            print(f"{name}: two thetas (the loop handles one): skipped")
            continue
        N = g["N"]
        M = (N + 63) // 64
        mpl = args.mpl or next(m for m in gen.MPLS if m * args.waves >= M)
        if args.targets:
            a, b = args.targets.split(":")
            targets = list(range(int(a), int(b)))
        else:
            targets = list(range(N))
        print(f"{name}: N={N} L={g['L']} W={len(g['wb']) - 1}")
        check_case(g["seq"], g["r"], g["wb"], g["theta_painter"], g["stones"], mpl, args.waves, targets)


if __name__ == "__main__":
    try:
        main()
    except EmuError as e:
        print("FAULT:", e)
        sys.exit(1)
