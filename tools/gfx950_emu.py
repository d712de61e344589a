#!/usr/bin/env python3
"""An interpreter for the gfx950 assembly the build keeps of its kernels (relate_amd/csrc/*.s): runs a compiled
kernel -- compiler-generated instructions and inline-asm statements alike -- wave by wave on the CPU.

Why: the register-resident kernels are hand-written instruction streams inside compiler-generated code; an error
in either is a GPU memory fault, which on the machines this library is developed on costs the whole box.  Here
every memory access (global, scalar, flat, scratch, LDS) is checked against the buffers the host would have
allocated, scalar loads are asynchronous until waited for (a register in flight may be neither read nor written),
and the records a kernel writes can be compared with the CPU oracle.

Scope: the instructions that occur in this library's kernels (about 170 mnemonics), wave64, IEEE arithmetic with
exact fused multiply-add; v_rcp_f64 / v_rsq / v_log / v_exp are taken as correctly rounded (the hardware's are
within an ulp), so results that pass through a division or a libm call may differ from the GPU's in the last bit.
Not modelled: timing, wait states (s_nop) and the data hazards they guard, vmcnt (vector loads complete at once).
"""
import math
import re
import struct
from fractions import Fraction

import numpy as np

U32 = 0xFFFFFFFF
U64 = 0xFFFFFFFFFFFFFFFF
# slots of the special registers in the scalar file
VCC, EXEC, M0_REG, SCRATCH_LO = 106, 126, 124, 102
LDS_APERTURE_HI = 0x00010000      # high dword of flat addresses that fall into LDS (src_shared_base)
SCRATCH_APERTURE_HI = 0x00020000  # ... into the private segment (src_private_base)


def _const_expr(text):
    """An assembler constant: a number (possibly negative), or sums and products of numbers (the assembler
    evaluates such expressions)."""
    text = text.strip()
    try:
        return int(text, 0)
    except ValueError:
        pass
    if re.fullmatch(r"[0-9a-fA-FxX+* ]+", text) is None:
        raise ValueError(f"not a constant expression: {text!r}")
    return sum(int(np.prod([int(f, 0) for f in term.split("*")])) for term in text.split("+"))


class EmuError(Exception):
    pass


class Memory:
    """Named buffers at addresses far apart; every access must lie inside one of them."""

    def __init__(self):
        self.bufs = []
        self.next = 0x7F00_0000_0000

    def add(self, name, data, alloc_bytes=None, writable=False):
        data = bytearray(data)
        n = len(data) if alloc_bytes is None else alloc_bytes
        if n < len(data):
            raise ValueError("allocation smaller than its contents")
        data += bytes(n - len(data))
        base = self.next
        self.next += ((n + 0xFFFF) // 0x10000 + 16) * 0x10000
        self.bufs.append([name, base, data, n, writable])
        return base

    def find(self, addr, nbytes, what):
        for b in self.bufs:
            if b[1] <= addr and addr + nbytes <= b[1] + b[3]:
                return b
        near = min(self.bufs, key=lambda b: min(abs(addr - b[1]), abs(addr - (b[1] + b[3]))))
        raise EmuError(f"{what}: {nbytes} bytes at {addr:#x} are outside every buffer "
                       f"(nearest: {near[0]} [{near[1]:#x}, {near[1] + near[3]:#x}), offset {addr - near[1]})")

    def read(self, addr, nbytes, what):
        b = self.find(addr, nbytes, what)
        return bytes(b[2][addr - b[1]:addr - b[1] + nbytes])

    def write(self, addr, data, what):
        b = self.find(addr, len(data), what)
        if not b[4]:
            raise EmuError(f"{what}: store into read-only buffer {b[0]}")
        b[2][addr - b[1]:addr - b[1] + len(data)] = data

    def buffer(self, name):
        for b in self.bufs:
            if b[0] == name:
                return b[2]
        raise KeyError(name)

    def address(self, name):
        for b in self.bufs:
            if b[0] == name:
                return b[1]
        raise KeyError(name)


# ---- exact arithmetic helpers ---------------------------------------------------------------------------------
def _f64(bits):
    return struct.unpack("<d", struct.pack("<Q", bits & U64))[0]


def _bits64(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def _f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits & U32))[0]


def _bits32(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def fma64(a, b, c):
    """round(a * b + c) with a single rounding"""
    if not (math.isfinite(a) and math.isfinite(b) and math.isfinite(c)):
        return a * b + c
    r = Fraction(a) * Fraction(b) + Fraction(c)
    if r == 0:
        # sign of an exact zero: + unless both addends are negative zeros
        return a * b + c
    try:
        return float(r)
    except OverflowError:
        return math.inf if r > 0 else -math.inf


def fma32(a, b, c):
    r = fma64(float(a), float(b), float(c))  # double has enough bits only for the product; go exact instead
    if not (math.isfinite(a) and math.isfinite(b) and math.isfinite(c)):
        return np.float32(r)
    fr = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    return np.float32(_round_fraction_to_f32(fr)) if fr != 0 else np.float32(r)


def _round_fraction_to_f32(fr):
    # float(Fraction) rounds to double correctly; a second rounding to float can double-round: do it exactly
    d = float(fr)
    f = np.float32(d)
    if not np.isfinite(f) or f == 0:
        return float(f)
    lo, hi = sorted((float(np.nextafter(f, np.float32(-np.inf))), float(np.nextafter(f, np.float32(np.inf)))))
    best = min((float(f), lo, hi), key=lambda c: (abs(Fraction(c) - fr), (_bits32(c) & 1)))
    return best


class Wave:
    def __init__(self, mem, lds, lds_bytes, scratch_bytes, name=""):
        self.s = np.zeros(128, dtype=np.uint64)
        self.v = np.zeros((512, 64), dtype=np.uint32)
        self.scc = 0
        self.mem, self.lds, self.lds_bytes = mem, lds, lds_bytes
        self.scratch = [bytearray(scratch_bytes) for _ in range(64)]
        self.scratch_bytes = scratch_bytes
        self.pending_s = {}
        self.gpr_idx = None
        self.name = name
        self.clock = 1000
        self.hw_id = 0
        self.stats = {}
        # registers no instruction has written yet hold whatever the previous wave left there: reads of them are
        # collected (the value here is zero, on the device it is not)
        self.sdef = np.zeros(128, dtype=bool)
        self.vdef = np.zeros(512, dtype=bool)
        self.undefined_reads = {}
        self.pc = -1
        self.line = ""
        self.set_s64(EXEC, U64)
        self.sdef[[VCC, VCC + 1, M0_REG]] = True

    # ---- scalar registers
    def get_s32(self, r):
        if r in self.pending_s:
            raise EmuError(f"{self.name}: s{r} is read before the scalar load that fills it has been waited for: {self.line}")
        if not self.sdef[r]:
            self.undefined_reads.setdefault((self.pc, self.line), set()).add(f"s{r}")
        return int(self.s[r]) & U32

    def set_s32(self, r, val):
        if r in self.pending_s:
            raise EmuError(f"{self.name}: s{r} is overwritten while a scalar load into it is in flight: {self.line}")
        self.s[r] = int(val) & U32
        self.sdef[r] = True

    def get_s64(self, r):
        return self.get_s32(r) | (self.get_s32(r + 1) << 32)

    def set_s64(self, r, val):
        self.set_s32(r, val & U32)
        self.set_s32(r + 1, (val >> 32) & U32)

    def exec_mask(self):
        m = self.get_s64(EXEC)
        return np.array([(m >> i) & 1 for i in range(64)], dtype=bool)

    # ---- operand parsing
    _sre = re.compile(r"^s(\d+)$|^s\[(\d+):(\d+)\]$")
    _vre = re.compile(r"^[va](\d+)$|^[va]\[(\d+):(\d+)\]$")  # a: accumulation registers, rows 256.. of the file
    _special = {"vcc": VCC, "vcc_lo": VCC, "vcc_hi": VCC + 1, "exec": EXEC, "exec_lo": EXEC, "exec_hi": EXEC + 1, "m0": M0_REG,
                "flat_scratch": SCRATCH_LO, "flat_scratch_lo": SCRATCH_LO, "flat_scratch_hi": SCRATCH_LO + 1}
    _fconst = {"0.5": 0.5, "-0.5": -0.5, "1.0": 1.0, "-1.0": -1.0, "2.0": 2.0, "-2.0": -2.0, "4.0": 4.0, "-4.0": -4.0,
               "0.15915494": 0.15915494309189532}

    def sreg(self, tok):
        tok = tok.strip()
        m = self._sre.match(tok)
        if m:
            return int(m.group(1) if m.group(1) is not None else m.group(2))
        return self._special.get(tok)

    def vreg(self, tok):
        tok = tok.strip()
        m = self._vre.match(tok)
        if not m:
            return None
        return int(m.group(1) if m.group(1) is not None else m.group(2)) + (256 if tok[0] == "a" else 0)

    def ssrc32(self, tok):
        tok = tok.strip()
        r = self.sreg(tok)
        if r is not None:
            return self.get_s32(r)
        if tok == "scc":
            return self.scc
        if tok == "src_shared_base":
            return 0
        if tok in self._fconst:
            return _bits32(self._fconst[tok])
        if tok == "execz":
            return int(self.get_s64(EXEC) == 0)
        if tok == "vccz":
            return int(self.get_s64(VCC) == 0)
        return int(tok, 0) & U32

    def ssrc64(self, tok):
        tok = tok.strip()
        r = self.sreg(tok)
        if r is not None:
            return self.get_s64(r)
        if tok == "src_shared_base":
            return LDS_APERTURE_HI << 32
        if tok == "src_private_base":
            return SCRATCH_APERTURE_HI << 32
        if tok in self._fconst:
            return _bits64(self._fconst[tok])
        v = int(tok, 0)
        return v & U64  # negative inline constants sign-extend; 32-bit literals zero-extend

    def vsrc32(self, tok):
        """32-bit vector source as uint32[64]"""
        tok = tok.strip()
        if tok.startswith("-") and tok[1:2] in ("v", "s", "|"):  # float negation of a register operand: the sign bit
            return self.vsrc32(tok[1:]) ^ np.uint32(0x80000000)
        if tok.startswith("|") and tok.endswith("|"):
            return self.vsrc32(tok[1:-1]) & np.uint32(0x7FFFFFFF)
        r = self.vreg(tok)
        if r is not None:
            self._vread(r, 1)
            return self.v[r].copy()
        return np.full(64, self.ssrc32(tok), dtype=np.uint32)

    def vsrc64(self, tok):
        tok = tok.strip()
        r = self.vreg(tok)
        if r is not None:
            self._vread(r, 2)
            return self.v[r].astype(np.uint64) | (self.v[r + 1].astype(np.uint64) << np.uint64(32))
        return np.full(64, self.ssrc64(tok), dtype=np.uint64)

    def vsrc_f64(self, tok):
        tok = tok.strip()
        neg = abs_ = False
        if tok.startswith("-"):
            if tok[1:] not in ("0.5", "1.0", "2.0", "4.0") and not tok[1:2].isdigit():
                neg, tok = True, tok[1:]
        if tok.startswith("|") and tok.endswith("|"):
            abs_, tok = True, tok[1:-1]
        r = self.vreg(tok)
        if r is not None:
            val = self.vsrc64(tok).view(np.float64).copy()
        else:
            sr = self.sreg(tok)
            if sr is not None:
                val = np.full(64, _f64(self.get_s64(sr)))
            elif tok in self._fconst:
                val = np.full(64, self._fconst[tok])
            else:
                lit = int(tok, 0)
                # a 32-bit literal of a 64-bit float operand supplies the high dword; small integers are integers
                val = np.full(64, float(lit) if -16 <= lit <= 64 else _f64((lit & U32) << 32))
        if abs_:
            val = np.abs(val)
        if neg:
            val = -val
        return val

    def vsrc_f32(self, tok):
        tok = tok.strip()
        neg = abs_ = False
        if tok.startswith("-") and not (tok[1:2].isdigit()):
            neg, tok = True, tok[1:]
        if tok.startswith("|") and tok.endswith("|"):
            abs_, tok = True, tok[1:-1]
        r = self.vreg(tok)
        if r is not None:
            self._vread(r, 1)
            val = self.v[r].view(np.float32).copy()
        elif tok in self._fconst:
            val = np.full(64, self._fconst[tok], dtype=np.float32)
        elif self.sreg(tok) is not None:
            val = np.full(64, _f32(self.get_s32(self.sreg(tok))), dtype=np.float32)
        else:
            lit = int(tok, 0)
            val = np.full(64, float(lit) if -16 <= lit <= 64 else _f32(lit), dtype=np.float32)
        if abs_:
            val = np.abs(val)
        if neg:
            val = -val
        return val

    def _vread(self, r, n):
        for i in range(n):
            if not self.vdef[r + i]:
                self.undefined_reads.setdefault((self.pc, self.line), set()).add(f"v{r + i}")

    def put32(self, tok, vals, mask=None):
        r = self.vreg(tok)
        if self.gpr_idx is not None:
            r += self.gpr_idx
        if mask is None:
            mask = self.exec_mask()
        self.v[r][mask] = np.asarray(vals).astype(np.uint32)[mask]
        self.vdef[r] = True

    def put64(self, tok, bits, mask=None):
        r = self.vreg(tok)
        bits = np.asarray(bits, dtype=np.uint64)
        if mask is None:
            mask = self.exec_mask()
        self.v[r][mask] = (bits & np.uint64(U32)).astype(np.uint32)[mask]
        self.v[r + 1][mask] = (bits >> np.uint64(32)).astype(np.uint32)[mask]
        self.vdef[r] = self.vdef[r + 1] = True

    def put_f64(self, tok, vals, mask=None):
        self.put64(tok, np.asarray(vals, dtype=np.float64).view(np.uint64), mask)

    def put_f32(self, tok, vals, mask=None):
        self.put32(tok, np.asarray(vals, dtype=np.float32).view(np.uint32), mask)

    def put_lanemask(self, tok, bools):
        """result of a vector compare: one bit per active lane, zero for inactive lanes"""
        ex = self.exec_mask()
        m = 0
        for l in range(64):
            if ex[l] and bools[l]:
                m |= 1 << l
        self.set_s64(self.sreg(tok), m)

    # ---- decode and execute
    def step(self, line):
        self.line = line
        op, _, rest = line.partition(" ")
        rest = rest.strip()
        self.stats[op] = self.stats.get(op, 0) + 1
        if "_dpp" in op or " quad_perm:" in rest or " row_" in rest and "row_mask" in rest:
            return self.op_dpp(op, rest)
        base = op
        for suf in ("_e32", "_e64"):
            if base.endswith(suf):
                base = base[:-4]
        args = split_operands(rest)
        f = getattr(self, "op_" + base, None)
        if f is None:
            raise EmuError(f"instruction not modelled: {line}")
        return f(*args)

    # ---------------------------------------------------------------- scalar ALU
    def _s2(self, d, a, b, fn, bits=32, scc="nz"):
        x = self.ssrc32(a) if bits == 32 else self.ssrc64(a)
        y = self.ssrc32(b) if bits == 32 else self.ssrc64(b)
        r = fn(x, y) & (U32 if bits == 32 else U64)
        (self.set_s32 if bits == 32 else self.set_s64)(self.sreg(d), r)
        if scc == "nz":
            self.scc = int(r != 0)

    def op_s_mov_b32(self, d, a):
        self.set_s32(self.sreg(d), self.ssrc32(a))

    def op_s_mov_b64(self, d, a):
        self.set_s64(self.sreg(d), self.ssrc64(a))

    def op_s_movk_i32(self, d, a):
        v = int(a, 0) & 0xFFFF
        self.set_s32(self.sreg(d), v - 0x10000 if v & 0x8000 else v)

    def op_s_addk_i32(self, d, a):
        v = int(a, 0) & 0xFFFF
        v = v - 0x10000 if v & 0x8000 else v
        r = self._i32(self.get_s32(self.sreg(d))) + v
        self.set_s32(self.sreg(d), r)
        self.scc = int(not -(1 << 31) <= r < (1 << 31))

    def op_s_mulk_i32(self, d, a):
        v = int(a, 0) & 0xFFFF
        v = v - 0x10000 if v & 0x8000 else v
        self.set_s32(self.sreg(d), self._i32(self.get_s32(self.sreg(d))) * v)

    def op_s_cmpk_eq_u32(self, a, b): self.scc = int(self.ssrc32(a) == (int(b, 0) & 0xFFFF))
    def op_s_cmpk_lg_u32(self, a, b): self.scc = int(self.ssrc32(a) != (int(b, 0) & 0xFFFF))
    def op_s_cmpk_gt_u32(self, a, b): self.scc = int(self.ssrc32(a) > (int(b, 0) & 0xFFFF))
    def op_s_cmpk_ge_u32(self, a, b): self.scc = int(self.ssrc32(a) >= (int(b, 0) & 0xFFFF))
    def op_s_cmpk_lt_u32(self, a, b): self.scc = int(self.ssrc32(a) < (int(b, 0) & 0xFFFF))
    def op_s_cmpk_le_u32(self, a, b): self.scc = int(self.ssrc32(a) <= (int(b, 0) & 0xFFFF))

    def _cmpk_i(self, a, b, fn):
        v = int(b, 0) & 0xFFFF
        v = v - 0x10000 if v & 0x8000 else v
        self.scc = int(fn(self._i32(self.ssrc32(a)), v))

    def op_s_cmpk_eq_i32(self, a, b): self._cmpk_i(a, b, lambda x, y: x == y)
    def op_s_cmpk_lg_i32(self, a, b): self._cmpk_i(a, b, lambda x, y: x != y)
    def op_s_cmpk_gt_i32(self, a, b): self._cmpk_i(a, b, lambda x, y: x > y)
    def op_s_cmpk_ge_i32(self, a, b): self._cmpk_i(a, b, lambda x, y: x >= y)
    def op_s_cmpk_lt_i32(self, a, b): self._cmpk_i(a, b, lambda x, y: x < y)
    def op_s_cmpk_le_i32(self, a, b): self._cmpk_i(a, b, lambda x, y: x <= y)

    def op_s_add_u32(self, d, a, b):
        r = self.ssrc32(a) + self.ssrc32(b)
        self.set_s32(self.sreg(d), r)
        self.scc = int(r > U32)

    def op_s_addc_u32(self, d, a, b):
        r = self.ssrc32(a) + self.ssrc32(b) + self.scc
        self.set_s32(self.sreg(d), r)
        self.scc = int(r > U32)

    def op_s_sub_u32(self, d, a, b):
        x, y = self.ssrc32(a), self.ssrc32(b)
        self.set_s32(self.sreg(d), x - y)
        self.scc = int(y > x)

    def op_s_subb_u32(self, d, a, b):
        x, y = self.ssrc32(a), self.ssrc32(b) + self.scc
        self.set_s32(self.sreg(d), x - y)
        self.scc = int(y > x)

    @staticmethod
    def _i32(x):
        x &= U32
        return x - (1 << 32) if x >> 31 else x

    @staticmethod
    def _i64(x):
        x &= U64
        return x - (1 << 64) if x >> 63 else x

    def op_s_add_i32(self, d, a, b):
        r = self._i32(self.ssrc32(a)) + self._i32(self.ssrc32(b))
        self.set_s32(self.sreg(d), r)
        self.scc = int(not -(1 << 31) <= r < (1 << 31))

    def op_s_sub_i32(self, d, a, b):
        r = self._i32(self.ssrc32(a)) - self._i32(self.ssrc32(b))
        self.set_s32(self.sreg(d), r)
        self.scc = int(not -(1 << 31) <= r < (1 << 31))

    def op_s_mul_i32(self, d, a, b):
        self.set_s32(self.sreg(d), self.ssrc32(a) * self.ssrc32(b))

    def op_s_mul_hi_u32(self, d, a, b):
        self.set_s32(self.sreg(d), (self.ssrc32(a) * self.ssrc32(b)) >> 32)

    def op_s_mul_hi_i32(self, d, a, b):
        self.set_s32(self.sreg(d), (self._i32(self.ssrc32(a)) * self._i32(self.ssrc32(b))) >> 32)

    def op_s_and_b32(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x & y)

    def op_s_or_b32(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x | y)

    def op_s_xor_b32(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x ^ y)

    def op_s_andn2_b32(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x & ~y)

    def op_s_and_b64(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x & y, 64)

    def op_s_or_b64(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x | y, 64)

    def op_s_xor_b64(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x ^ y, 64)

    def op_s_andn2_b64(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x & ~y, 64)

    def op_s_orn2_b64(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x | ~y, 64)

    def op_s_not_b32(self, d, a):
        r = ~self.ssrc32(a) & U32
        self.set_s32(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_not_b64(self, d, a):
        r = ~self.ssrc64(a) & U64
        self.set_s64(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_lshl_b32(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x << (y & 31))

    def op_s_lshr_b32(self, d, a, b):
        self._s2(d, a, b, lambda x, y: x >> (y & 31))

    def op_s_lshl_b64(self, d, a, b):
        r = (self.ssrc64(a) << (self.ssrc32(b) & 63)) & U64
        self.set_s64(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_lshr_b64(self, d, a, b):
        r = self.ssrc64(a) >> (self.ssrc32(b) & 63)
        self.set_s64(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_ashr_i32(self, d, a, b):
        r = (self._i32(self.ssrc32(a)) >> (self.ssrc32(b) & 31)) & U32
        self.set_s32(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_ashr_i64(self, d, a, b):
        r = (self._i64(self.ssrc64(a)) >> (self.ssrc32(b) & 63)) & U64
        self.set_s64(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_bcnt1_i32_b32(self, d, a):
        r = bin(self.ssrc32(a)).count("1")
        self.set_s32(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_bcnt1_i32_b64(self, d, a):
        r = bin(self.ssrc64(a)).count("1")
        self.set_s32(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_ff1_i32_b64(self, d, a):
        x = self.ssrc64(a)
        self.set_s32(self.sreg(d), (x & -x).bit_length() - 1 if x else U32)

    def op_s_ff1_i32_b32(self, d, a):
        x = self.ssrc32(a)
        self.set_s32(self.sreg(d), (x & -x).bit_length() - 1 if x else U32)

    def op_s_min_i32(self, d, a, b):
        x, y = self._i32(self.ssrc32(a)), self._i32(self.ssrc32(b))
        self.set_s32(self.sreg(d), min(x, y))
        self.scc = int(x <= y)

    def op_s_max_i32(self, d, a, b):
        x, y = self._i32(self.ssrc32(a)), self._i32(self.ssrc32(b))
        self.set_s32(self.sreg(d), max(x, y))
        self.scc = int(x >= y)

    def op_s_min_u32(self, d, a, b):
        x, y = self.ssrc32(a), self.ssrc32(b)
        self.set_s32(self.sreg(d), min(x, y))
        self.scc = int(x <= y)

    def op_s_max_u32(self, d, a, b):
        x, y = self.ssrc32(a), self.ssrc32(b)
        self.set_s32(self.sreg(d), max(x, y))
        self.scc = int(x >= y)

    def op_s_cselect_b32(self, d, a, b):
        self.set_s32(self.sreg(d), self.ssrc32(a) if self.scc else self.ssrc32(b))

    def op_s_cselect_b64(self, d, a, b):
        self.set_s64(self.sreg(d), self.ssrc64(a) if self.scc else self.ssrc64(b))

    def op_s_bfe_u32(self, d, a, b):
        x, c = self.ssrc32(a), self.ssrc32(b)
        off, width = c & 31, (c >> 16) & 0x7F
        r = (x >> off) & ((1 << width) - 1)
        self.set_s32(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_bfe_i64(self, d, a, b):
        x, c = self.ssrc64(a), self.ssrc32(b)
        off, width = c & 63, (c >> 16) & 0x7F
        r = (x >> off) & ((1 << width) - 1)
        if width and r >> (width - 1):
            r -= 1 << width
        self.set_s64(self.sreg(d), r & U64)
        self.scc = int(r != 0)

    def op_s_bfe_u64(self, d, a, b):
        x, c = self.ssrc64(a), self.ssrc32(b)
        off, width = c & 63, (c >> 16) & 0x7F
        r = (x >> off) & ((1 << width) - 1)
        self.set_s64(self.sreg(d), r)
        self.scc = int(r != 0)

    def op_s_sext_i32_i16(self, d, a):
        v = self.ssrc32(a) & 0xFFFF
        self.set_s32(self.sreg(d), v - 0x10000 if v & 0x8000 else v)

    def op_s_bitcmp1_b32(self, a, b):
        self.scc = (self.ssrc32(a) >> (self.ssrc32(b) & 31)) & 1

    def op_s_bitcmp0_b32(self, a, b):
        self.scc = 1 - ((self.ssrc32(a) >> (self.ssrc32(b) & 31)) & 1)

    def _scmp(self, a, b, fn, signed=False, bits=32):
        if bits == 64:
            x, y = self.ssrc64(a), self.ssrc64(b)
        else:
            x, y = self.ssrc32(a), self.ssrc32(b)
            if signed:
                x, y = self._i32(x), self._i32(y)
        self.scc = int(fn(x, y))

    def op_s_cmp_eq_u32(self, a, b): self._scmp(a, b, lambda x, y: x == y)
    def op_s_cmp_lg_u32(self, a, b): self._scmp(a, b, lambda x, y: x != y)
    def op_s_cmp_gt_u32(self, a, b): self._scmp(a, b, lambda x, y: x > y)
    def op_s_cmp_ge_u32(self, a, b): self._scmp(a, b, lambda x, y: x >= y)
    def op_s_cmp_lt_u32(self, a, b): self._scmp(a, b, lambda x, y: x < y)
    def op_s_cmp_le_u32(self, a, b): self._scmp(a, b, lambda x, y: x <= y)
    def op_s_cmp_eq_i32(self, a, b): self._scmp(a, b, lambda x, y: x == y, True)
    def op_s_cmp_lg_i32(self, a, b): self._scmp(a, b, lambda x, y: x != y, True)
    def op_s_cmp_gt_i32(self, a, b): self._scmp(a, b, lambda x, y: x > y, True)
    def op_s_cmp_ge_i32(self, a, b): self._scmp(a, b, lambda x, y: x >= y, True)
    def op_s_cmp_lt_i32(self, a, b): self._scmp(a, b, lambda x, y: x < y, True)
    def op_s_cmp_le_i32(self, a, b): self._scmp(a, b, lambda x, y: x <= y, True)
    def op_s_cmp_eq_u64(self, a, b): self._scmp(a, b, lambda x, y: x == y, bits=64)
    def op_s_cmp_lg_u64(self, a, b): self._scmp(a, b, lambda x, y: x != y, bits=64)

    def op_s_and_saveexec_b64(self, d, a):
        old = self.get_s64(EXEC)
        new = self.ssrc64(a) & old
        self.set_s64(self.sreg(d), old)
        self.set_s64(EXEC, new)
        self.scc = int(new != 0)

    def op_s_or_saveexec_b64(self, d, a):
        old = self.get_s64(EXEC)
        new = self.ssrc64(a) | old
        self.set_s64(self.sreg(d), old)
        self.set_s64(EXEC, new)
        self.scc = int(new != 0)

    def op_s_andn2_saveexec_b64(self, d, a):
        old = self.get_s64(EXEC)
        new = self.ssrc64(a) & ~old & U64
        self.set_s64(self.sreg(d), old)
        self.set_s64(EXEC, new)
        self.scc = int(new != 0)

    def op_s_getreg_b32(self, d, what):
        # HW_ID / XCC_ID: where the wave runs; any value will do
        self.set_s32(self.sreg(d), self.hw_id if "HW_ID" in what and "XCC" not in what else 0)

    def op_s_setreg_b32(self, *a):
        pass

    def op_s_setreg_imm32_b32(self, *a):
        pass

    def op_s_memrealtime(self, d):
        self.clock += 100
        self.pending_s[self.sreg(d)] = self.clock & U32
        self.pending_s[self.sreg(d) + 1] = 0

    op_s_memtime = op_s_memrealtime

    def op_s_cbranch_scc0(self, label): return ("branch", label) if not self.scc else None
    def op_s_cbranch_scc1(self, label): return ("branch", label) if self.scc else None
    def op_s_cbranch_vccz(self, label): return ("branch", label) if self.get_s64(VCC) == 0 else None
    def op_s_cbranch_vccnz(self, label): return ("branch", label) if self.get_s64(VCC) != 0 else None
    def op_s_cbranch_execz(self, label): return ("branch", label) if self.get_s64(EXEC) == 0 else None
    def op_s_cbranch_execnz(self, label): return ("branch", label) if self.get_s64(EXEC) != 0 else None
    def op_s_branch(self, label): return ("branch", label)
    def op_s_endpgm(self): return ("end",)
    def op_s_barrier(self): return ("barrier",)
    def op_s_setprio(self, a): pass
    def op_s_nop(self, a): pass
    def op_s_sleep(self, a): pass
    def op_s_trap(self, a): raise EmuError(f"{self.name}: s_trap {a} reached")

    def op_s_waitcnt(self, *what):
        text = " ".join(what)
        if "lgkmcnt(0)" in text:
            for r, val in self.pending_s.items():
                self.s[r] = val
                self.sdef[r] = True
            self.pending_s.clear()

    def op_s_set_gpr_idx_on(self, a, mode):
        if int(mode, 0) != 8:
            raise EmuError("only destination-relative index mode is modelled")
        self.gpr_idx = self.ssrc32(a) & 0xFF
        self.s[M0_REG] = (int(self.s[M0_REG]) & ~0xF0FF) | self.gpr_idx | (8 << 12)

    def op_s_set_gpr_idx_off(self):
        self.gpr_idx = None

    def _sload(self, d, base, off, ndw):
        r0 = self.sreg(d)
        off = off.strip()
        o = self.get_s32(self.sreg(off)) if self.sreg(off) is not None else int(off, 0)
        addr = (self.ssrc64(base) + o) & ~3
        data = self.mem.read(addr, 4 * ndw, f"{self.name}: s_load_dwordx{ndw} into s[{r0}:{r0 + ndw - 1}] ({self.line})")
        for i in range(ndw):
            if r0 + i in self.pending_s:
                raise EmuError(f"{self.name}: two scalar loads in flight into s{r0 + i}")
            self.pending_s[r0 + i] = struct.unpack_from("<I", data, 4 * i)[0]

    def op_s_load_dword(self, d, base, off): self._sload(d, base, off, 1)
    def op_s_load_dwordx2(self, d, base, off): self._sload(d, base, off, 2)
    def op_s_load_dwordx4(self, d, base, off): self._sload(d, base, off, 4)
    def op_s_load_dwordx8(self, d, base, off): self._sload(d, base, off, 8)
    def op_s_load_dwordx16(self, d, base, off): self._sload(d, base, off, 16)

    # ---------------------------------------------------------------- vector ALU: moves, integers
    def op_v_mov_b32(self, d, a):
        self.put32(d, self.vsrc32(a))

    def op_v_mov_b64(self, d, a):
        self.put64(d, self.vsrc64(a))

    def op_v_readlane_b32(self, d, a, lane):
        self.set_s32(self.sreg(d), int(self.v[self.vreg(a)][self.ssrc32(lane) & 63]))

    def op_v_readfirstlane_b32(self, d, a):
        m = self.exec_mask()
        lane = int(np.argmax(m)) if m.any() else 0
        self.set_s32(self.sreg(d), int(self.v[self.vreg(a)][lane]))

    def op_v_writelane_b32(self, d, a, lane):
        self.v[self.vreg(d)][self.ssrc32(lane) & 63] = self.ssrc32(a)
        self.vdef[self.vreg(d)] = True

    def _vi(self, d, fn, *srcs):
        vals = [self.vsrc32(s).astype(np.uint64) for s in srcs]
        self.put32(d, (fn(*vals) & np.uint64(U32)))

    def op_v_add_u32(self, d, a, b): self._vi(d, lambda x, y: x + y, a, b)
    def op_v_sub_u32(self, d, a, b): self._vi(d, lambda x, y: x - y + np.uint64(1 << 32), a, b)
    def op_v_subrev_u32(self, d, a, b): self._vi(d, lambda x, y: y - x + np.uint64(1 << 32), a, b)
    def op_v_add3_u32(self, d, a, b, c): self._vi(d, lambda x, y, z: x + y + z, a, b, c)
    def op_v_and_b32(self, d, a, b): self._vi(d, lambda x, y: x & y, a, b)
    def op_v_or_b32(self, d, a, b): self._vi(d, lambda x, y: x | y, a, b)
    def op_v_xor_b32(self, d, a, b): self._vi(d, lambda x, y: x ^ y, a, b)
    def op_v_not_b32(self, d, a): self._vi(d, lambda x: ~x, a)
    def op_v_and_or_b32(self, d, a, b, c): self._vi(d, lambda x, y, z: (x & y) | z, a, b, c)
    def op_v_or3_b32(self, d, a, b, c): self._vi(d, lambda x, y, z: x | y | z, a, b, c)
    def op_v_lshlrev_b32(self, d, a, b): self._vi(d, lambda x, y: y << (x & np.uint64(31)), a, b)
    def op_v_lshrrev_b32(self, d, a, b): self._vi(d, lambda x, y: y >> (x & np.uint64(31)), a, b)
    def op_v_lshl_add_u32(self, d, a, b, c): self._vi(d, lambda x, y, z: (x << (y & np.uint64(31))) + z, a, b, c)
    def op_v_lshl_or_b32(self, d, a, b, c): self._vi(d, lambda x, y, z: (x << (y & np.uint64(31))) | z, a, b, c)
    def op_v_add_lshl_u32(self, d, a, b, c): self._vi(d, lambda x, y, z: (x + y) << (z & np.uint64(31)), a, b, c)
    def op_v_mul_lo_u32(self, d, a, b): self._vi(d, lambda x, y: x * y, a, b)
    def op_v_mul_hi_u32(self, d, a, b): self._vi(d, lambda x, y: (x * y) >> np.uint64(32), a, b)
    def op_v_mul_u32_u24(self, d, a, b): self._vi(d, lambda x, y: (x & np.uint64(0xFFFFFF)) * (y & np.uint64(0xFFFFFF)), a, b)
    def op_v_bfe_u32(self, d, a, b, c):
        self._vi(d, lambda x, y, z: (x >> (y & np.uint64(31))) & ((np.uint64(1) << (z & np.uint64(31))) - np.uint64(1)), a, b, c)

    def op_v_ashrrev_i32(self, d, a, b):
        sh = self.vsrc32(a) & 31
        self.put32(d, (self.vsrc32(b).view(np.int32) >> sh.astype(np.int32)).view(np.uint32))

    def op_v_mad_u32_u24(self, d, a, b, c):
        self._vi(d, lambda x, y, z: (x & np.uint64(0xFFFFFF)) * (y & np.uint64(0xFFFFFF)) + z, a, b, c)

    def op_v_mad_u64_u32(self, d, carry, a, b, c):
        x, y, z = self.vsrc32(a).astype(object), self.vsrc32(b).astype(object), self.vsrc64(c).astype(object)
        r = x * y + z
        self.put_lanemask(carry, np.array([int(v) > U64 for v in r]))
        self.put64(d, np.array([int(v) & U64 for v in r], dtype=np.uint64))

    def op_v_mad_i64_i32(self, d, carry, a, b, c):
        x = self.vsrc32(a).view(np.int32).astype(object)
        y = self.vsrc32(b).view(np.int32).astype(object)
        z = self.vsrc64(c).view(np.int64).astype(object)
        r = x * y + z
        self.put_lanemask(carry, np.array([not -(1 << 63) <= int(v) < (1 << 63) for v in r]))
        self.put64(d, np.array([int(v) & U64 for v in r], dtype=np.uint64))

    def op_v_bcnt_u32_b32(self, d, a, b):
        x = self.vsrc32(a)
        self.put32(d, np.array([bin(int(v)).count("1") for v in x], dtype=np.uint32) + self.vsrc32(b))

    def op_v_ffbl_b32(self, d, a):
        x = self.vsrc32(a)
        self.put32(d, np.array([((int(v) & -int(v)).bit_length() - 1) & U32 if v else U32 for v in x], dtype=np.uint32))

    def op_v_ffbh_u32(self, d, a):
        x = self.vsrc32(a)
        self.put32(d, np.array([32 - int(v).bit_length() if v else U32 for v in x], dtype=np.uint32))

    def op_v_min_u32(self, d, a, b): self.put32(d, np.minimum(self.vsrc32(a), self.vsrc32(b)))
    def op_v_max_u32(self, d, a, b): self.put32(d, np.maximum(self.vsrc32(a), self.vsrc32(b)))
    def op_v_min_i32(self, d, a, b): self.put32(d, np.minimum(self.vsrc32(a).view(np.int32), self.vsrc32(b).view(np.int32)).view(np.uint32))
    def op_v_max_i32(self, d, a, b): self.put32(d, np.maximum(self.vsrc32(a).view(np.int32), self.vsrc32(b).view(np.int32)).view(np.uint32))
    def op_v_min_f32(self, d, a, b): self.put_f32(d, np.fmin(self.vsrc_f32(a), self.vsrc_f32(b)))
    def op_v_max_f32(self, d, a, b): self.put_f32(d, np.fmax(self.vsrc_f32(a), self.vsrc_f32(b)))

    def op_v_xad_u32(self, d, a, b, c): self._vi(d, lambda x, y, z: (x ^ y) + z, a, b, c)

    def op_v_mul_i32_i24(self, d, a, b):
        def s24(x):
            x = x.astype(np.int64) & 0xFFFFFF
            return np.where(x & 0x800000, x - 0x1000000, x)
        self.put32(d, ((s24(self.vsrc32(a)) * s24(self.vsrc32(b))) & U32).astype(np.uint32))

    def op_v_bfrev_b32(self, d, a):
        self.put32(d, np.array([int(f"{int(v):032b}"[::-1], 2) for v in self.vsrc32(a)], dtype=np.uint32))

    def op_v_ashrrev_i64(self, d, a, b):
        sh = (self.vsrc32(a) & 63).astype(np.int64)
        self.put64(d, (self.vsrc64(b).view(np.int64) >> sh).view(np.uint64))

    def op_v_cmp_ne_u16(self, dst, a, b):
        self.put_lanemask(dst, (self.vsrc32(a) & 0xFFFF) != (self.vsrc32(b) & 0xFFFF))

    def op_v_cmp_eq_u16(self, dst, a, b):
        self.put_lanemask(dst, (self.vsrc32(a) & 0xFFFF) == (self.vsrc32(b) & 0xFFFF))

    def op_v_add_u32_sdwa(self, d, a, b_and_mods):
        parts = b_and_mods.split()
        sel = {p.split(":")[0]: p.split(":")[1] for p in parts[1:]}
        if sel.get("dst_sel", "DWORD") != "DWORD":
            raise EmuError("SDWA destination selection not modelled: " + self.line)

        def pick(x, how):
            if how == "DWORD":
                return x
            kind, idx = how.split("_")
            if kind == "BYTE":
                return (x >> np.uint32(8 * int(idx))) & np.uint32(0xFF)
            if kind == "WORD":
                return (x >> np.uint32(16 * int(idx))) & np.uint32(0xFFFF)
            raise EmuError("SDWA selection not modelled: " + how)
        x = pick(self.vsrc32(a), sel.get("src0_sel", "DWORD")).astype(np.uint64)
        y = pick(self.vsrc32(parts[0]), sel.get("src1_sel", "DWORD")).astype(np.uint64)
        self.put32(d, (x + y) & np.uint64(U32))

    def op_v_fmamk_f32(self, d, a, k, b):
        self.op_v_fma_f32(d, a, k, b)

    def op_v_fmaak_f32(self, d, a, b, k):
        self.op_v_fma_f32(d, a, b, k)

    def op_v_cvt_u32_f32(self, d, a):
        x = np.nan_to_num(np.trunc(self.vsrc_f32(a).astype(np.float64)), nan=0.0, posinf=2**32 - 1, neginf=0)
        self.put32(d, np.clip(x, 0, 2**32 - 1).astype(np.uint64).astype(np.uint32))

    def op_v_cvt_i32_f32(self, d, a):
        x = np.nan_to_num(np.trunc(self.vsrc_f32(a).astype(np.float64)), nan=0.0, posinf=2**31 - 1, neginf=-2**31)
        self.put32(d, np.clip(x, -2**31, 2**31 - 1).astype(np.int64).astype(np.int32).view(np.uint32))

    def op_v_trunc_f32(self, d, a):
        self.put_f32(d, np.trunc(self.vsrc_f32(a)))

    def op_v_rndne_f32(self, d, a):
        self.put_f32(d, np.rint(self.vsrc_f32(a)))

    def op_v_rcp_iflag_f32(self, d, a):
        self.op_v_rcp_f32(d, a)

    def _pk_f32(self, d, a, b_and_mods, fn):
        parts = b_and_mods.split()
        mods = {"op_sel": [0, 0], "op_sel_hi": [1, 1], "neg_lo": [0, 0], "neg_hi": [0, 0]}
        for p in parts[1:]:
            k, _, v = p.partition(":")
            mods[k] = [int(x) for x in v.strip("[]").split(",")]

        def halves(tok):
            tok = tok.strip()
            r = self.vreg(tok)
            if r is not None:
                self._vread(r, 2)
                return self.v[r].view(np.float32).copy(), self.v[r + 1].view(np.float32).copy()
            sr = self.sreg(tok)
            if sr is not None:
                return (np.full(64, _f32(self.get_s32(sr)), dtype=np.float32), np.full(64, _f32(self.get_s32(sr + 1)), dtype=np.float32))
            c = self._fconst[tok] if tok in self._fconst else _f32(int(tok, 0))
            return np.full(64, c, dtype=np.float32), np.zeros(64, dtype=np.float32)  # a constant sits in the low half
        x, y = halves(a), halves(parts[0])
        out = []
        for sel, neg in ((mods["op_sel"], mods["neg_lo"]), (mods["op_sel_hi"], mods["neg_hi"])):
            p, q = x[sel[0]], y[sel[1]]
            p = -p if neg[0] else p
            q = -q if neg[1] else q
            with np.errstate(all="ignore"):
                out.append(fn(p, q).astype(np.float32))
        r = self.vreg(d)
        m = self.exec_mask()
        self.v[r][m] = out[0].view(np.uint32)[m]
        self.v[r + 1][m] = out[1].view(np.uint32)[m]
        self.vdef[r] = self.vdef[r + 1] = True

    def op_v_pk_add_f32(self, d, a, b): self._pk_f32(d, a, b, lambda p, q: p + q)
    def op_v_pk_mul_f32(self, d, a, b): self._pk_f32(d, a, b, lambda p, q: p * q)

    def op_v_alignbit_b32(self, d, a, b, c):
        x = (self.vsrc32(a).astype(np.uint64) << np.uint64(32)) | self.vsrc32(b).astype(np.uint64)
        self.put32(d, (x >> (self.vsrc32(c).astype(np.uint64) & np.uint64(31))) & np.uint64(U32))

    def op_v_lshl_add_u64(self, d, a, b, c):
        x, sh, z = self.vsrc64(a), self.vsrc32(b).astype(np.uint64) & np.uint64(7), self.vsrc64(c)
        with np.errstate(over="ignore"):
            self.put64(d, (x << sh) + z)

    def op_v_lshlrev_b64(self, d, a, b):
        sh = self.vsrc32(a).astype(np.uint64) & np.uint64(63)
        self.put64(d, self.vsrc64(b) << sh)

    def op_v_lshrrev_b64(self, d, a, b):
        sh = self.vsrc32(a).astype(np.uint64) & np.uint64(63)
        self.put64(d, self.vsrc64(b) >> sh)

    def op_v_add_co_u32(self, d, carry, a, b):
        r = self.vsrc32(a).astype(np.uint64) + self.vsrc32(b).astype(np.uint64)
        self.put_lanemask(carry, r > U32)
        self.put32(d, r & np.uint64(U32))

    def op_v_sub_co_u32(self, d, carry, a, b):
        x, y = self.vsrc32(a).astype(np.uint64), self.vsrc32(b).astype(np.uint64)
        self.put_lanemask(carry, y > x)
        self.put32(d, (x - y + np.uint64(1 << 32)) & np.uint64(U32))

    def op_v_subrev_co_u32(self, d, carry, a, b):
        self.op_v_sub_co_u32(d, carry, b, a)

    def _lanemask_bits(self, tok):
        m = self.ssrc64(tok)
        return np.array([(m >> i) & 1 for i in range(64)], dtype=np.uint64)

    def op_v_addc_co_u32(self, d, carry, a, b, cin):
        c = self._lanemask_bits(cin)
        r = self.vsrc32(a).astype(np.uint64) + self.vsrc32(b).astype(np.uint64) + c
        self.put_lanemask(carry, r > U32)
        self.put32(d, r & np.uint64(U32))

    def op_v_subb_co_u32(self, d, carry, a, b, cin):
        c = self._lanemask_bits(cin)
        x, y = self.vsrc32(a).astype(np.uint64), self.vsrc32(b).astype(np.uint64) + c
        self.put_lanemask(carry, y > x)
        self.put32(d, (x - y + np.uint64(1 << 33)) & np.uint64(U32))

    def op_v_subbrev_co_u32(self, d, carry, a, b, cin):
        self.op_v_subb_co_u32(d, carry, b, a, cin)

    def op_v_cndmask_b32(self, d, a, b, sel="vcc"):
        m = self._lanemask_bits(sel).astype(bool)
        self.put32(d, np.where(m, self.vsrc32(b), self.vsrc32(a)))

    def op_v_mbcnt_lo_u32_b32(self, d, a, b):
        m = self.vsrc32(a)
        lanes = np.arange(64)
        cnt = np.array([bin(int(m[l]) & ((1 << min(l, 32)) - 1)).count("1") for l in lanes], dtype=np.uint32)
        self.put32(d, cnt + self.vsrc32(b))

    def op_v_mbcnt_hi_u32_b32(self, d, a, b):
        m = self.vsrc32(a)
        cnt = np.array([bin(int(m[l]) & ((1 << max(l - 32, 0)) - 1)).count("1") for l in range(64)], dtype=np.uint32)
        self.put32(d, cnt + self.vsrc32(b))

    # ---- vector compares: *_e32 writes VCC, *_e64 an SGPR pair
    def _vcmp(self, args, fn, kind):
        if len(args) == 3:
            dst, a, b = args
        else:
            dst, (a, b) = "vcc", args
        if kind == "u32":
            x, y = self.vsrc32(a).astype(np.int64), self.vsrc32(b).astype(np.int64)
        elif kind == "i32":
            x, y = self.vsrc32(a).view(np.int32).astype(np.int64), self.vsrc32(b).view(np.int32).astype(np.int64)
        elif kind == "u64":
            x, y = self.vsrc64(a), self.vsrc64(b)
        elif kind == "i64":
            x, y = self.vsrc64(a).view(np.int64), self.vsrc64(b).view(np.int64)
        elif kind == "f64":
            x, y = self.vsrc_f64(a), self.vsrc_f64(b)
        else:
            x, y = self.vsrc_f32(a), self.vsrc_f32(b)
        with np.errstate(invalid="ignore"):
            self.put_lanemask(dst, fn(x, y))

    def op_v_cmp_class_f64(self, dst, a, mask):
        x = self.vsrc_f64(a)
        m = self.vsrc32(mask)
        bits = x.view(np.uint64)
        out = np.zeros(64, dtype=bool)
        for l in range(64):
            v, b = float(x[l]), int(bits[l])
            neg = b >> 63
            e, f = (b >> 52) & 0x7FF, b & ((1 << 52) - 1)
            if e == 0x7FF and f:
                c = 1 if (f >> 51) else 0  # bit 0 signalling, bit 1 quiet
                c = 1 - c
            elif e == 0x7FF:
                c = 2 if neg else 9
            elif e == 0 and f == 0:
                c = 5 if neg else 6
            elif e == 0:
                c = 4 if neg else 7
            else:
                c = 3 if neg else 8
            out[l] = (int(m[l]) >> c) & 1
            del v
        self.put_lanemask(dst, out)

    # ---------------------------------------------------------------- vector ALU: floating point
    def op_v_add_f64(self, d, a, b):
        with np.errstate(all="ignore"):
            self.put_f64(d, self.vsrc_f64(a) + self.vsrc_f64(b))

    def op_v_mul_f64(self, d, a, b):
        with np.errstate(all="ignore"):
            self.put_f64(d, self.vsrc_f64(a) * self.vsrc_f64(b))

    def op_v_max_f64(self, d, a, b):
        self.put_f64(d, np.fmax(self.vsrc_f64(a), self.vsrc_f64(b)))

    def op_v_min_f64(self, d, a, b):
        self.put_f64(d, np.fmin(self.vsrc_f64(a), self.vsrc_f64(b)))

    def op_v_fma_f64(self, d, a, b, c):
        x, y, z = self.vsrc_f64(a), self.vsrc_f64(b), self.vsrc_f64(c)
        ex = self.exec_mask()
        out = np.zeros(64)
        for l in range(64):
            if ex[l]:
                out[l] = fma64(float(x[l]), float(y[l]), float(z[l]))
        self.put_f64(d, out)

    def op_v_fmac_f64(self, d, a, b):
        self.op_v_fma_f64(d, a, b, d)

    def op_v_add_f32(self, d, a, b):
        with np.errstate(all="ignore"):
            self.put_f32(d, self.vsrc_f32(a) + self.vsrc_f32(b))

    def op_v_sub_f32(self, d, a, b):
        with np.errstate(all="ignore"):
            self.put_f32(d, self.vsrc_f32(a) - self.vsrc_f32(b))

    def op_v_subrev_f32(self, d, a, b):
        with np.errstate(all="ignore"):
            self.put_f32(d, self.vsrc_f32(b) - self.vsrc_f32(a))

    def op_v_mul_f32(self, d, a, b):
        with np.errstate(all="ignore"):
            self.put_f32(d, self.vsrc_f32(a) * self.vsrc_f32(b))

    def op_v_fma_f32(self, d, a, b, c):
        x, y, z = self.vsrc_f32(a), self.vsrc_f32(b), self.vsrc_f32(c)
        ex = self.exec_mask()
        out = np.zeros(64, dtype=np.float32)
        for l in range(64):
            if ex[l]:
                out[l] = fma32(x[l], y[l], z[l])
        self.put_f32(d, out)

    def op_v_fmac_f32(self, d, a, b):
        self.op_v_fma_f32(d, a, b, d)

    def op_v_cvt_f32_f64(self, d, a):
        with np.errstate(all="ignore"):
            self.put_f32(d, self.vsrc_f64(a).astype(np.float32))

    def op_v_cvt_f64_f32(self, d, a):
        self.put_f64(d, self.vsrc_f32(a).astype(np.float64))

    def op_v_cvt_f64_i32(self, d, a):
        self.put_f64(d, self.vsrc32(a).view(np.int32).astype(np.float64))

    def op_v_cvt_f64_u32(self, d, a):
        self.put_f64(d, self.vsrc32(a).astype(np.float64))

    def op_v_cvt_f32_i32(self, d, a):
        self.put_f32(d, self.vsrc32(a).view(np.int32).astype(np.float32))

    def op_v_cvt_f32_u32(self, d, a):
        self.put_f32(d, self.vsrc32(a).astype(np.float32))

    def op_v_cvt_i32_f64(self, d, a):
        x = np.nan_to_num(np.trunc(self.vsrc_f64(a)), nan=0.0, posinf=2**31 - 1, neginf=-2**31)
        self.put32(d, np.clip(x, -2**31, 2**31 - 1).astype(np.int64).astype(np.int32).view(np.uint32))

    def op_v_cvt_u32_f64(self, d, a):
        x = np.nan_to_num(np.trunc(self.vsrc_f64(a)), nan=0.0, posinf=2**32 - 1, neginf=0)
        self.put32(d, np.clip(x, 0, 2**32 - 1).astype(np.uint64).astype(np.uint32))

    def op_v_rcp_f64(self, d, a):
        with np.errstate(all="ignore"):
            self.put_f64(d, 1.0 / self.vsrc_f64(a))

    def op_v_rcp_f32(self, d, a):
        with np.errstate(all="ignore"):
            self.put_f32(d, np.float32(1.0) / self.vsrc_f32(a))

    def op_v_ldexp_f64(self, d, a, b):
        x = self.vsrc_f64(a)
        e = self.vsrc32(b).view(np.int32)
        with np.errstate(all="ignore"):
            self.put_f64(d, np.ldexp(x, e))

    def op_v_frexp_mant_f64(self, d, a):
        x = self.vsrc_f64(a)
        m, _ = np.frexp(x)
        m = np.where(np.isfinite(x), m, x)
        self.put_f64(d, m)

    def op_v_frexp_exp_i32_f64(self, d, a):
        x = self.vsrc_f64(a)
        _, e = np.frexp(x)
        e = np.where(np.isfinite(x) & (x != 0), e, 0)
        self.put32(d, e.astype(np.int32).view(np.uint32))

    def op_v_rndne_f64(self, d, a):
        self.put_f64(d, np.rint(self.vsrc_f64(a)))

    def op_v_trunc_f64(self, d, a):
        self.put_f64(d, np.trunc(self.vsrc_f64(a)))

    def op_v_div_scale_f64(self, d, vcc_out, s0, s1, s2):
        """Scaling of numerator / denominator before the division sequence.  For operands whose quotient and
        reciprocal stay well inside the normal range (all this library divides) nothing is scaled."""
        a, b, c = self.vsrc_f64(s0), self.vsrc_f64(s1), self.vsrc_f64(s2)
        ok = np.isfinite(b) & np.isfinite(c) & (b != 0) & (c != 0)
        with np.errstate(all="ignore"):
            eb = np.frexp(b)[1]
            ec = np.frexp(c)[1]
        if not np.all(~self.exec_mask() | (ok & (np.abs(ec - eb) < 900) & (np.abs(eb) < 900) & (np.abs(ec) < 900))):
            raise EmuError(f"{self.name}: v_div_scale_f64 on operands that need scaling (not modelled): {self.line}")
        self.put_lanemask(vcc_out, np.zeros(64, dtype=bool))
        self.put_f64(d, a)

    def op_v_div_fmas_f64(self, d, a, b, c):
        self.op_v_fma_f64(d, a, b, c)  # VCC is clear (no scaling): a plain fused multiply-add

    def op_v_div_fixup_f64(self, d, q, den, num):
        # special cases aside (none occur: see v_div_scale_f64), the quotient passes through
        self.put_f64(d, self.vsrc_f64(q))

    def op_v_sqrt_f64(self, d, a):
        with np.errstate(all="ignore"):
            self.put_f64(d, np.sqrt(self.vsrc_f64(a)))

    # ---------------------------------------------------------------- DPP
    def op_dpp(self, op, rest):
        m = re.match(r"(v\d+),\s*(v\d+)\s+(.*?)\s+row_mask:(\w+)\s+bank_mask:(\w+)(\s+bound_ctrl:\S+)?$", rest)
        if not m or not op.startswith("v_mov_b32"):
            raise EmuError("DPP form not modelled: " + op + " " + rest)
        d, a, ctrl, row_mask, bank_mask = self.vreg(m.group(1)), self.vreg(m.group(2)), m.group(3), int(m.group(4), 0), int(m.group(5), 0)
        bound = m.group(6) is not None
        src, out = self.v[a].copy(), self.v[d].copy()
        ex = self.exec_mask()
        for lane in range(64):
            row, inrow = lane >> 4, lane & 15
            if not ex[lane] or not (row_mask >> row) & 1 or not (bank_mask >> (inrow >> 2)) & 1:
                continue
            sl = None
            if ctrl.startswith("quad_perm:"):
                p = [int(x) for x in ctrl[len("quad_perm:["):-1].split(",")]
                sl = (lane & ~3) | p[lane & 3]
            elif ctrl == "row_half_mirror":
                sl = (lane & ~7) | (7 - (lane & 7))
            elif ctrl == "row_mirror":
                sl = (lane & ~15) | (15 - inrow)
            elif ctrl.startswith("row_shr:"):
                k = int(ctrl[8:])
                sl = lane - k if inrow >= k else None
            elif ctrl.startswith("row_shl:"):
                k = int(ctrl[8:])
                sl = lane + k if inrow + k < 16 else None
            elif ctrl.startswith("row_ror:"):
                k = int(ctrl[8:])
                sl = (lane & ~15) | ((inrow - k) & 15)
            elif ctrl == "row_bcast:15":
                sl = 16 * (row - 1) + 15 if row > 0 else None
            elif ctrl == "row_bcast:31":
                sl = 31 if row >= 2 else None
            elif ctrl == "wave_shr:1":
                sl = lane - 1 if lane > 0 else None
            else:
                raise EmuError("DPP control not modelled: " + ctrl)
            if sl is None or not ex[sl]:
                if bound:
                    out[lane] = 0
                continue
            out[lane] = src[sl]
        self._vread(a, 1)
        self.v[d] = out
        self.vdef[d] = True

    # ---------------------------------------------------------------- memory
    @staticmethod
    def _mods(tokens):
        off = 0
        for t in tokens:
            for part in t.split():
                if part.startswith("offset:"):
                    off = _const_expr(part[7:])
        return off

    def _gaddr(self, vaddr, saddr, off, lane):
        vaddr, saddr = vaddr.strip(), saddr.strip()
        vr = self.vreg(vaddr)
        if saddr == "off":  # 64-bit address in a VGPR pair
            base = int(self.v[vr][lane]) | (int(self.v[vr + 1][lane]) << 32)
        else:
            base = self.ssrc64(saddr) + int(self.v[vr][lane])
        return (base + off) & U64

    def _vload(self, kind, d, vaddr, saddr, mods, ndw):
        off = self._mods(mods)
        ex = self.exec_mask()
        r = self.vreg(d)
        for lane in range(64):
            if not ex[lane]:
                continue
            addr = self._gaddr(vaddr, saddr, off, lane)
            data = self._flat_read(addr, 4 * ndw, lane, f"{self.name}: {kind} lane {lane} ({self.line})")
            for i in range(ndw):
                self.v[r + i][lane] = struct.unpack_from("<I", data, 4 * i)[0]
        self.vdef[r:r + ndw] = True

    def _vstore(self, kind, vaddr, data, saddr, mods, ndw):
        off = self._mods(mods)
        ex = self.exec_mask()
        r = self.vreg(data)
        for lane in range(64):
            if not ex[lane]:
                continue
            addr = self._gaddr(vaddr, saddr, off, lane)
            raw = b"".join(struct.pack("<I", int(self.v[r + i][lane])) for i in range(ndw))
            self._flat_write(addr, raw, lane, f"{self.name}: {kind} lane {lane} ({self.line})")

    def _flat_read(self, addr, n, lane, what):
        hi = addr >> 32
        if hi == LDS_APERTURE_HI:
            a = addr & U32
            if a + n > self.lds_bytes:
                raise EmuError(f"{what}: LDS access at {a} beyond the allocation of {self.lds_bytes} bytes")
            return bytes(self.lds[a:a + n])
        if hi == SCRATCH_APERTURE_HI:
            a = addr & U32
            if a + n > self.scratch_bytes:
                raise EmuError(f"{what}: scratch access at {a} beyond the private segment of {self.scratch_bytes} bytes")
            return bytes(self.scratch[lane][a:a + n])
        return self.mem.read(addr, n, what)

    def _flat_write(self, addr, raw, lane, what):
        hi = addr >> 32
        n = len(raw)
        if hi == LDS_APERTURE_HI:
            a = addr & U32
            if a + n > self.lds_bytes:
                raise EmuError(f"{what}: LDS access at {a} beyond the allocation of {self.lds_bytes} bytes")
            self.lds[a:a + n] = raw
        elif hi == SCRATCH_APERTURE_HI:
            a = addr & U32
            if a + n > self.scratch_bytes:
                raise EmuError(f"{what}: scratch access at {a} beyond the private segment of {self.scratch_bytes} bytes")
            self.scratch[lane][a:a + n] = raw
        else:
            self.mem.write(addr, raw, what)

    def op_global_load_dword(self, d, vaddr, saddr, *m): self._vload("global_load_dword", d, vaddr, saddr.split()[0], (saddr,) + m, 1)
    def op_global_load_dwordx2(self, d, vaddr, saddr, *m): self._vload("global_load_dwordx2", d, vaddr, saddr.split()[0], (saddr,) + m, 2)
    def op_global_load_dwordx3(self, d, vaddr, saddr, *m): self._vload("global_load_dwordx3", d, vaddr, saddr.split()[0], (saddr,) + m, 3)
    def op_global_load_dwordx4(self, d, vaddr, saddr, *m): self._vload("global_load_dwordx4", d, vaddr, saddr.split()[0], (saddr,) + m, 4)
    def op_global_store_dword(self, vaddr, data, saddr, *m): self._vstore("global_store_dword", vaddr, data, saddr.split()[0], (saddr,) + m, 1)
    def op_global_store_dwordx2(self, vaddr, data, saddr, *m): self._vstore("global_store_dwordx2", vaddr, data, saddr.split()[0], (saddr,) + m, 2)
    def op_global_store_dwordx3(self, vaddr, data, saddr, *m): self._vstore("global_store_dwordx3", vaddr, data, saddr.split()[0], (saddr,) + m, 3)
    def op_global_store_dwordx4(self, vaddr, data, saddr, *m): self._vstore("global_store_dwordx4", vaddr, data, saddr.split()[0], (saddr,) + m, 4)

    def op_global_load_ubyte(self, d, vaddr, saddr, *m):
        off = self._mods((saddr,) + m)
        ex = self.exec_mask()
        for lane in range(64):
            if ex[lane]:
                a = self._gaddr(vaddr, saddr.split()[0], off, lane)
                self.v[self.vreg(d)][lane] = self._flat_read(a, 1, lane, f"{self.name}: global_load_ubyte ({self.line})")[0]

    def op_global_store_byte(self, vaddr, data, saddr, *m):
        off = self._mods((saddr,) + m)
        ex = self.exec_mask()
        for lane in range(64):
            if ex[lane]:
                a = self._gaddr(vaddr, saddr.split()[0], off, lane)
                self._flat_write(a, bytes([int(self.v[self.vreg(data)][lane]) & 0xFF]), lane, f"{self.name}: global_store_byte ({self.line})")

    def op_flat_load_dword(self, d, vaddr, *m): self._vload("flat_load_dword", d, vaddr.split()[0], "off", (vaddr,) + m, 1)
    def op_flat_load_dwordx2(self, d, vaddr, *m): self._vload("flat_load_dwordx2", d, vaddr.split()[0], "off", (vaddr,) + m, 2)
    def op_flat_store_dword(self, vaddr, data, *m): self._vstore("flat_store_dword", vaddr, data.split()[0], "off", (data,) + m, 1)
    def op_flat_store_dwordx2(self, vaddr, data, *m): self._vstore("flat_store_dwordx2", vaddr, data.split()[0], "off", (data,) + m, 2)

    def _scratch_addr(self, vaddr, saddr, off, lane):
        a = off
        if vaddr.strip() != "off":
            a += int(self.v[self.vreg(vaddr)][lane])
        if saddr.strip() != "off":
            a += self.ssrc32(saddr)
        return a

    def _scratch(self, load, reg, vaddr, saddr, mods, ndw):
        off = self._mods(mods)
        ex = self.exec_mask()
        r = self.vreg(reg)
        for lane in range(64):
            if not ex[lane]:
                continue
            a = self._scratch_addr(vaddr, saddr.split()[0], off, lane)
            if a < 0 or a + 4 * ndw > self.scratch_bytes:
                raise EmuError(f"{self.name}: scratch access at {a} beyond the private segment of {self.scratch_bytes} bytes ({self.line})")
            for i in range(ndw):
                if load:
                    self.vdef[r + i] = True
                    self.v[r + i][lane] = struct.unpack_from("<I", self.scratch[lane], a + 4 * i)[0]
                else:
                    struct.pack_into("<I", self.scratch[lane], a + 4 * i, int(self.v[r + i][lane]))

    def op_scratch_load_dword(self, d, vaddr, saddr, *m): self._scratch(True, d, vaddr, saddr, (saddr,) + m, 1)
    def op_scratch_load_dwordx2(self, d, vaddr, saddr, *m): self._scratch(True, d, vaddr, saddr, (saddr,) + m, 2)
    def op_scratch_load_dwordx3(self, d, vaddr, saddr, *m): self._scratch(True, d, vaddr, saddr, (saddr,) + m, 3)
    def op_scratch_load_dwordx4(self, d, vaddr, saddr, *m): self._scratch(True, d, vaddr, saddr, (saddr,) + m, 4)
    def op_scratch_store_dword(self, vaddr, data, saddr, *m): self._scratch(False, data, vaddr, saddr, (saddr,) + m, 1)
    def op_scratch_store_dwordx2(self, vaddr, data, saddr, *m): self._scratch(False, data, vaddr, saddr, (saddr,) + m, 2)
    def op_scratch_store_dwordx3(self, vaddr, data, saddr, *m): self._scratch(False, data, vaddr, saddr, (saddr,) + m, 3)
    def op_scratch_store_dwordx4(self, vaddr, data, saddr, *m): self._scratch(False, data, vaddr, saddr, (saddr,) + m, 4)

    # LDS
    def _lds_check(self, a, n):
        if a < 0 or a + n > self.lds_bytes:
            raise EmuError(f"{self.name}: LDS access of {n} bytes at {a} (allocation {self.lds_bytes}): {self.line}")

    def _ds_write(self, addr, datas, offs, width):
        ex = self.exec_mask()
        for lane in range(64):
            if not ex[lane]:
                continue
            base = int(self.v[self.vreg(addr.split()[0])][lane])
            for data, off in zip(datas, offs):
                r = self.vreg(data.split()[0])
                a = base + off
                self._lds_check(a, 4 * width)
                for i in range(width):
                    struct.pack_into("<I", self.lds, a + 4 * i, int(self.v[r + i][lane]))

    def _ds_read(self, d, addr, offs, width):
        ex = self.exec_mask()
        r = self.vreg(d)
        for lane in range(64):
            if not ex[lane]:
                continue
            base = int(self.v[self.vreg(addr.split()[0])][lane])
            for k, off in enumerate(offs):
                a = base + off
                self._lds_check(a, 4 * width)
                for i in range(width):
                    self.v[r + k * width + i][lane] = struct.unpack_from("<I", self.lds, a + 4 * i)[0]
        self.vdef[r:r + len(offs) * width] = True

    @staticmethod
    def _off2(tokens, scale):
        o0 = o1 = 0
        for t in tokens:
            for part in t.split():
                if part.startswith("offset0:"):
                    o0 = int(part[8:], 0)
                if part.startswith("offset1:"):
                    o1 = int(part[8:], 0)
        return [o0 * scale, o1 * scale]

    def op_ds_write_b32(self, addr, data, *m): self._ds_write(addr, [data], [self._mods((data,) + m)], 1)
    def op_ds_write_b64(self, addr, data, *m): self._ds_write(addr, [data], [self._mods((data,) + m)], 2)
    def op_ds_write_b128(self, addr, data, *m): self._ds_write(addr, [data], [self._mods((data,) + m)], 4)
    def op_ds_write2_b32(self, addr, d0, d1, *m): self._ds_write(addr, [d0, d1], self._off2((d1,) + m, 4), 1)
    def op_ds_write2_b64(self, addr, d0, d1, *m): self._ds_write(addr, [d0, d1], self._off2((d1,) + m, 8), 2)
    def op_ds_write2st64_b32(self, addr, d0, d1, *m): self._ds_write(addr, [d0, d1], self._off2((d1,) + m, 256), 1)
    def op_ds_read_b32(self, d, addr, *m): self._ds_read(d, addr, [self._mods((addr,) + m)], 1)
    def op_ds_read_b64(self, d, addr, *m): self._ds_read(d, addr, [self._mods((addr,) + m)], 2)
    def op_ds_read_b128(self, d, addr, *m): self._ds_read(d, addr, [self._mods((addr,) + m)], 4)
    def op_ds_read2_b32(self, d, addr, *m): self._ds_read(d, addr, self._off2((addr,) + m, 4), 1)
    def op_ds_read2_b64(self, d, addr, *m): self._ds_read(d, addr, self._off2((addr,) + m, 8), 2)
    def op_ds_read2st64_b32(self, d, addr, *m): self._ds_read(d, addr, self._off2((addr,) + m, 256), 1)
    def op_ds_read2st64_b64(self, d, addr, *m): self._ds_read(d, addr, self._off2((addr,) + m, 512), 2)

    def op_ds_bpermute_b32(self, d, addr, data, *m):
        """every lane fetches `data` of lane addr / 4 (mod 64); a source lane that is switched off delivers 0"""
        ex = self.exec_mask()
        off = self._mods((data,) + m)
        a = self.vsrc32(addr.split()[0])
        src = self.vsrc32(data.split()[0])
        out = np.zeros(64, dtype=np.uint32)
        for lane in range(64):
            sl = ((int(a[lane]) + off) >> 2) & 63
            out[lane] = src[sl] if ex[sl] else 0
        self.put32(d, out)

    def op_ds_swizzle_b32(self, d, data, *m):
        """lane permutation inside the LDS crossbar (no memory access).  offset[15] = 1: quad-permute mode, offset[7:0]
        holds the source lane of each lane of a quad; offset[15] = 0: bit-mask mode inside groups of 32 lanes,
        source lane = ((lane & and_mask) | or_mask) ^ xor_mask with the masks in offset[4:0], [9:5], [14:10].  A
        source lane that is switched off delivers 0."""
        ex = self.exec_mask()
        off = self._mods((data,) + m)
        src = self.vsrc32(data.split()[0])
        out = np.zeros(64, dtype=np.uint32)
        for lane in range(64):
            if off & 0x8000:
                sl = (lane & ~3) | ((off >> (2 * (lane & 3))) & 3)
            else:
                sl = (lane & 32) | ((((lane & 31) & (off & 31)) | ((off >> 5) & 31)) ^ ((off >> 10) & 31))
            out[lane] = src[sl] if ex[sl] else 0
        self.put32(d, out)

    # accumulation registers: rows 256.. of the vector register file here (the hardware's unified file)
    @staticmethod
    def _areg(tok):
        m = re.fullmatch(r"a\[([^\]:]+)\]|a(\d+)", tok.strip())
        if not m:
            raise ValueError(f"not an accumulation register: {tok!r}")
        n = _const_expr(m.group(1)) if m.group(1) is not None else int(m.group(2))
        if not 0 <= n < 256:
            raise ValueError(f"accumulation register out of range: {tok!r}")
        return 256 + n

    def op_v_accvgpr_write_b32(self, d, s):
        r = self._areg(d)
        vals = self.vsrc32(s)
        ex = self.exec_mask()
        self.v[r][ex] = np.asarray(vals, dtype=np.uint32)[ex] if np.ndim(vals) else np.uint32(vals)
        self.vdef[r] = True

    def op_v_accvgpr_mov_b32(self, d, s):
        r, q = self._areg(d), self._areg(s)
        if not self.vdef[q]:
            raise EmuError(f"{self.name}: accumulation register read before anything wrote it: {self.line}")
        ex = self.exec_mask()
        self.v[r][ex] = self.v[q][ex]
        self.vdef[r] = True

    def op_v_accvgpr_read_b32(self, d, s):
        r = self._areg(s)
        if not self.vdef[r]:
            raise EmuError(f"{self.name}: accumulation register read before anything wrote it: {self.line}")
        self.put32(d, self.v[r].copy())

    def op_ds_permute_b32(self, d, addr, data, *m):
        ex = self.exec_mask()
        off = self._mods((data,) + m)
        a = self.vsrc32(addr.split()[0])
        src = self.vsrc32(data.split()[0])
        out = np.zeros(64, dtype=np.uint32)
        for lane in range(64):
            if ex[lane]:
                out[((int(a[lane]) + off) >> 2) & 63] = src[lane]
        self.put32(d, out)

    def op_buffer_wbl2(self, *a): pass
    def op_buffer_inv(self, *a): pass
    def op_s_dcache_wb(self, *a): pass
    def op_s_icache_inv(self, *a): pass


def _make_cmps():
    ops = {"eq": lambda x, y: x == y, "ne": lambda x, y: x != y, "lg": lambda x, y: x != y, "gt": lambda x, y: x > y,
           "ge": lambda x, y: x >= y, "lt": lambda x, y: x < y, "le": lambda x, y: x <= y,
           # the negated float compares are true for unordered operands
           "neq": lambda x, y: ~(x == y), "nlg": lambda x, y: ~(x != y) | np.isnan(x) | np.isnan(y),
           "ngt": lambda x, y: ~(x > y), "nge": lambda x, y: ~(x >= y), "nlt": lambda x, y: ~(x < y), "nle": lambda x, y: ~(x <= y),
           "o": lambda x, y: ~(np.isnan(x) | np.isnan(y)), "u": lambda x, y: np.isnan(x) | np.isnan(y)}
    for name, fn in ops.items():
        for kind in ("u32", "i32", "u64", "i64", "f64", "f32"):
            if name in ("neq", "nlg", "ngt", "nge", "nlt", "nle", "o", "u") and kind[0] != "f":
                continue
            if name == "lg" and kind[0] != "f":
                continue

            def method(self, *args, _fn=fn, _kind=kind):
                self._vcmp(args, _fn, _kind)
            setattr(Wave, f"op_v_cmp_{name}_{kind}", method)


_make_cmps()


def split_operands(rest):
    """operands separated by commas outside brackets; modifiers stay attached to the last operand"""
    out, depth, cur = [], 0, ""
    for ch in rest:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


_BR = re.compile(r"\b([vsa])\[([0-9a-fA-Fx+\-* ]+)(?::([0-9a-fA-Fx+\-* ]+))?\]")


def _fold_brackets(line):
    """inline asm writes register numbers as expressions (s[40:40+1], v[37]): fold them to plain numbers"""
    def sub(m):
        lo = eval(m.group(2), {"__builtins__": {}})
        if m.group(3) is None:
            return f"{m.group(1)}{lo}"
        hi = eval(m.group(3), {"__builtins__": {}})
        return f"{m.group(1)}[{lo}:{hi}]"
    return _BR.sub(sub, line)


# ---- kernels from an assembly file ------------------------------------------------------------------------------
class Kernel:
    def __init__(self, name, lines, labels, meta):
        self.name, self.lines, self.labels, self.meta = name, lines, labels, meta


def load_kernel(path, name_pattern):
    """The kernel whose mangled name matches `name_pattern` (a regular expression) from a .s file."""
    rx = re.compile(name_pattern)
    lines, labels, meta = [], {}, {}
    name = None
    with open(path) as f:
        for raw in f:
            if name is None:
                m = re.match(r"^(\w+):", raw)
                if m and rx.search(m.group(1)) and not raw.startswith(".L"):
                    name = m.group(1)
                continue
            code = raw.split(";")[0].rstrip()
            if not code.strip():
                continue
            s = code.strip()
            if s.startswith(".amdhsa_"):
                k, _, v = s.partition(" ")
                meta[k[8:]] = v.strip()
                continue
            if s.startswith(".end_amdhsa_kernel"):
                break
            m = re.match(r"^([.\w$]+):$", s)
            if m:
                labels[m.group(1)] = len(lines)
                continue
            if s.startswith("."):
                continue
            lines.append(_fold_brackets(s))
    if name is None:
        raise KeyError(name_pattern)
    return Kernel(name, lines, labels, meta)


def kernel_args_layout(path, kernel_name):
    """The argument list of a kernel from the code-object metadata at the end of the .s file: a list of
    (offset, size, value_kind)."""
    import yaml
    text = open(path).read()
    a = text.index(".amdgpu_metadata")
    b = text.rindex("\n", 0, text.index(".end_amdgpu_metadata"))
    meta = yaml.safe_load(text[a + len(".amdgpu_metadata"):b])
    for k in meta["amdhsa.kernels"]:
        if k[".name"] == kernel_name:
            return [(x[".offset"], x[".size"], x[".value_kind"]) for x in k.get(".args", [])], k
    raise KeyError(kernel_name)


def pack_kernarg(layout, explicit, grid, block):
    """Kernel-argument block: `explicit` (bytes objects, one per explicit argument, in order) at the offsets of
    the metadata, the hidden arguments the device runtime would add (grid and workgroup sizes) after them."""
    size = max(o + n for o, n, _ in layout) if layout else 0
    buf = bytearray(size)
    it = iter(explicit)
    for off, n, kind in layout:
        if not kind.startswith("hidden_"):
            data = next(it)
            if len(data) != n:
                raise ValueError(f"argument at offset {off} takes {n} bytes, got {len(data)}")
            buf[off:off + n] = data
        elif kind.startswith("hidden_block_count_"):
            buf[off:off + 4] = struct.pack("<I", grid["xyz".index(kind[-1])])
        elif kind.startswith("hidden_group_size_"):
            buf[off:off + 2] = struct.pack("<H", block["xyz".index(kind[-1])])
        elif kind.startswith("hidden_remainder_"):
            buf[off:off + 2] = struct.pack("<H", 0)
        elif kind == "hidden_grid_dims":
            buf[off:off + 2] = struct.pack("<H", 1 + (grid[1] > 1) + (grid[2] > 1))
    if next(it, None) is not None:
        raise ValueError("more arguments than the kernel takes")
    return bytes(buf)


def launch(path, name_pattern, mem, explicit, grid, block, lds_bytes=0, garbage=None):
    """kernel<<<grid, block, lds_bytes>>>(explicit...) in the interpreter; grid and block are (x, y, z), the
    workgroup is one-dimensional.  Returns the merged instruction statistics."""
    kern = load_kernel(path, name_pattern)
    layout, _ = kernel_args_layout(path, kern.name)
    ka = pack_kernarg(layout, explicit, grid, block)
    kernarg = mem.add("kernarg:" + kern.name[-24:], ka, max(len(ka), 8))
    if block[1] != 1 or block[2] != 1:
        raise ValueError("one-dimensional workgroups only")
    stats = {}
    n_waves = (block[0] + 63) // 64
    for bz in range(grid[2]):
        for by in range(grid[1]):
            for bx in range(grid[0]):
                waves = run_workgroup(kern, mem, kernarg, (bx, by, bz), n_waves, lds_bytes, garbage=garbage, threads=block[0])
                for w in waves:
                    for k, v in w.stats.items():
                        stats[k] = stats.get(k, 0) + v
    return stats


def run_workgroup(kernel, mem, kernarg_addr, block_id, n_waves, lds_bytes, budget=200_000_000, on_step=None, garbage=None,
                  threads=None):
    """Runs one workgroup of `n_waves` waves to completion; returns the waves (for their statistics).
    garbage: a seed; registers, LDS and scratch then start with random contents instead of zeros, as on the
    device, where they hold what the previous wave left (a kernel whose result or addresses depend on that fails)."""
    scratch = int(kernel.meta.get("private_segment_fixed_size", "0"))
    lds = bytearray(max(lds_bytes, int(kernel.meta.get("group_segment_fixed_size", "0"))))
    rng = np.random.default_rng(garbage) if garbage is not None else None
    if rng is not None:
        lds[:] = rng.integers(0, 256, len(lds), dtype=np.uint8).tobytes()
    waves = []
    for g in range(n_waves):
        w = Wave(mem, lds, len(lds), scratch, f"{kernel.name[:48]} block {block_id} wave {g}")
        if rng is not None:
            w.s[:102] = rng.integers(0, 1 << 32, 102, dtype=np.uint64)
            w.s[VCC:VCC + 2] = rng.integers(0, 1 << 32, 2, dtype=np.uint64)
            w.s[M0_REG] = int(rng.integers(0, 1 << 32))
            w.v[:] = rng.integers(0, 1 << 32, w.v.shape, dtype=np.uint64).astype(np.uint32)
            w.scc = int(rng.integers(0, 2))
            for l in range(64):
                w.scratch[l][:] = rng.integers(0, 256, scratch, dtype=np.uint8).tobytes()
        # user SGPRs, then the system SGPRs the descriptor enables
        ids = block_id if isinstance(block_id, tuple) else (block_id, 0, 0)
        meta = kernel.meta
        r = 0
        for key, n in (("user_sgpr_dispatch_ptr", 2), ("user_sgpr_queue_ptr", 2)):
            if meta.get(key, "0") == "1":
                r += n  # not used by this library's kernels: left undefined
        if meta.get("user_sgpr_kernarg_segment_ptr", "1") == "1":
            w.set_s64(r, kernarg_addr)
            r += 2
        if meta.get("user_sgpr_dispatch_id", "0") == "1":
            r += 2
        for axis, key in enumerate(("system_sgpr_workgroup_id_x", "system_sgpr_workgroup_id_y", "system_sgpr_workgroup_id_z")):
            if meta.get(key, "1" if axis == 0 else "0") == "1":
                w.set_s32(r, ids[axis])
                r += 1
        w.v[0] = (np.arange(64) + 64 * g).astype(np.uint32)
        w.vdef[0] = True
        if threads is not None and threads < 64 * (g + 1):
            live = max(0, threads - 64 * g)
            w.set_s64(EXEC, (1 << live) - 1)
        w.hw_id = g
        waves.append(w)
    pcs = [0] * n_waves
    done = [False] * n_waves
    while not all(done):
        at_barrier = 0
        for wi, w in enumerate(waves):
            if done[wi]:
                continue
            while True:
                budget -= 1
                if budget < 0:
                    raise EmuError("instruction budget exhausted")
                if pcs[wi] >= len(kernel.lines):
                    raise EmuError(f"{w.name}: ran past the end of the kernel")
                line = kernel.lines[pcs[wi]]
                pcs[wi] += 1
                w.pc = pcs[wi] - 1
                if on_step:
                    on_step(w, pcs[wi] - 1, line)
                try:
                    res = w.step(line)
                except EmuError:
                    raise
                except Exception as e:
                    raise EmuError(f"{w.name}: interpreter failed at instruction {pcs[wi] - 1} `{line}`: {e!r}") from e
                if res is None:
                    continue
                if res[0] == "branch":
                    if res[1] not in kernel.labels:
                        raise EmuError(f"branch to unknown label {res[1]}")
                    pcs[wi] = kernel.labels[res[1]]
                elif res[0] == "barrier":
                    at_barrier += 1
                    break
                elif res[0] == "end":
                    done[wi] = True
                    break
        if at_barrier and at_barrier != sum(1 for x in done if not x):
            raise EmuError("some waves of the workgroup finished while others wait at a barrier")
    return waves
