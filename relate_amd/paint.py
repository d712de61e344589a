"""Host-side mirror of the reference's painting interface over the C ABI.

  paint_chunk   <-> int Paint(cxxopts::ParseResult&, int chunk_index)   pipeline/Paint.cpp:15-91
  paint_stones  <-> the loop over FastPainting::PaintSteppingStones     pipeline/Paint.cpp:78-82
  Data          <-> Data(6 files), Data::SetPainting                    src/data.cpp:88-99,:124-129
  FastPainting  <-> FastPainting(N, theta), PaintSteppingStones(basename, ...)  src/fast_painting.cpp:639-652
  Plan          <-> device-resident plan (inputs/outputs in HBM), used by the benchmark and the multi-GPU host
"""
import ctypes as C

import numpy as np

from ._lib import check, f32p, f64p, i32p, i64p, lib, record_bytes, u8p, vp


def _wb(wb):
    return np.ascontiguousarray(wb, dtype=np.int32)


def paint_stones(seq, r, theta_painter, theta_data, wb, k_begin=0, k_end=None, device=0):
    """Records of targets [k_begin,k_end) as a (W, K, 56+8N) uint8 array (window-major, as in relate_<w>.bin)."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    r = np.ascontiguousarray(r, dtype=np.float64)
    wb = _wb(wb)
    L, N = seq.shape
    W = len(wb) - 1
    k_end = N if k_end is None else k_end
    out = np.zeros((W, max(k_end - k_begin, 0), record_bytes(N)), dtype=np.uint8)
    check(lib().relate_paint_stones(N, L, seq.ctypes.data_as(u8p), r.ctypes.data_as(f64p), float(theta_painter),
                                    float(theta_data), W, wb.ctypes.data_as(i32p), k_begin, k_end,
                                    out.ctypes.data_as(u8p), device))
    return out


def paint_chunk(output_dir, chunk_index, theta=None, rho=None, device=-1):
    """Drop-in for the Paint stage: reads <output>/chunk_<c>.*, writes <output>/chunk_<c>/paint/relate_<w>.bin."""
    t = C.c_double(theta) if theta is not None else None
    rh = C.c_double(rho) if rho is not None else None
    check(lib().relate_paint_chunk(str(output_dir).encode(), chunk_index, C.byref(t) if t is not None else None,
                                   C.byref(rh) if rh is not None else None, device))


class Data:
    """Chunk data as the painting stage sees it (src/data.hpp:29-42)."""

    def __init__(self, hap, pos, dist, rec, rpos, state, Ne=30000, mu=1.25e-8):
        self._h = vp()
        enc = lambda s: None if s is None else str(s).encode()  # noqa: E731
        check(lib().relate_data_new(enc(hap), enc(pos), enc(dist), enc(rec), enc(rpos), enc(state), Ne, mu,
                                    C.byref(self._h)))
        n, l = C.c_int(), C.c_int()
        check(lib().relate_data_dims(self._h, C.byref(n), C.byref(l)))
        self.N, self.L = n.value, l.value

    def SetPainting(self, theta, rho):
        check(lib().relate_data_set_painting(self._h, float(theta), float(rho)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().relate_data_free(self._h)
            self._h = None


class FastPainting:
    """FastPainting(N, theta) as the Rust caller constructs it (src/pipelines.rs:181)."""

    def __init__(self, N, theta=0.001, device=0):
        self._h = vp()
        check(lib().relate_painter_new(N, float(theta), device, C.byref(self._h)))

    def PaintSteppingStones(self, data, basename, window_boundaries, k):
        wb = _wb(window_boundaries)
        check(lib().relate_painter_paint_stepping_stones(self._h, data._h, str(basename).encode(), len(wb) - 1,
                                                         wb.ctypes.data_as(i32p), k))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().relate_painter_free(self._h)
            self._h = None


class Plan:
    """Device-resident painting plan.  Device pointers are plain integers (e.g. torch.Tensor.data_ptr())."""

    def __init__(self, device=0, variant=1, waves_per_target=0):
        self._h = vp()
        check(lib().relate_plan_create(device, C.byref(self._h)))
        check(lib().relate_plan_configure(self._h, variant, waves_per_target))
        self.N = self.L = self.W = self.K = 0

    def close(self):
        if getattr(self, "_h", None):
            lib().relate_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def _set(self, N, L, wb):
        self.N, self.L, self.W, self.K = N, L, len(wb) - 1, 0

    def load_device(self, N, L, d_seq, d_r, wb, stream=0):
        wb = _wb(wb)
        check(lib().relate_plan_load_device(self._h, N, L, vp(d_seq), vp(d_r), len(wb) - 1, wb.ctypes.data_as(i32p),
                                            vp(stream)))
        self._set(N, L, wb)

    def load_device_into(self, N, L, d_seq, d_r, wb, d_packed, nbytes, stream=0):
        """Packs into a caller-owned device buffer (e.g. a torch tensor that is then broadcast)."""
        wb = _wb(wb)
        check(lib().relate_plan_load_device_into(self._h, N, L, vp(d_seq), vp(d_r), len(wb) - 1,
                                                 wb.ctypes.data_as(i32p), vp(d_packed), nbytes, vp(stream)))
        self._set(N, L, wb)

    def load_host(self, seq, r, wb):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        r = np.ascontiguousarray(r, dtype=np.float64)
        wb = _wb(wb)
        L, N = seq.shape
        check(lib().relate_plan_load_host(self._h, N, L, seq.ctypes.data_as(u8p), r.ctypes.data_as(f64p), len(wb) - 1,
                                          wb.ctypes.data_as(i32p)))
        self._set(N, L, wb)

    def load_rows_host(self, N, rows, r, wb, rho=1.0):
        """Bit rows (L x relate_row_stride_words(N) uint64, bits from N on set) and r from host arrays."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        r = np.ascontiguousarray(r, dtype=np.float64)
        wb = _wb(wb)
        L = rows.shape[0]
        if rows.ndim != 2 or rows.shape[1] != lib().relate_row_stride_words(N) or r.shape != (L,):
            raise ValueError("rows must be (L, relate_row_stride_words(N)) and r (L,)")
        check(lib().relate_plan_load_rows_host(self._h, N, L, rows.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               r.ctypes.data_as(f64p), float(rho), len(wb) - 1, wb.ctypes.data_as(i32p)))
        self._set(N, L, wb)

    def packed(self):
        p, n = vp(), C.c_size_t()
        check(lib().relate_plan_packed(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def load_packed(self, N, L, d_packed, nbytes, d_r, wb, stream=0):
        wb = _wb(wb)
        check(lib().relate_plan_load_packed(self._h, N, L, vp(d_packed), nbytes, vp(d_r), len(wb) - 1,
                                            wb.ctypes.data_as(i32p), vp(stream)))
        self._set(N, L, wb)

    def prepare(self, theta_painter, theta_data, k_begin, k_end, stream=0):
        s = C.c_int64()
        check(lib().relate_plan_prepare(self._h, float(theta_painter), float(theta_data), k_begin, k_end, C.byref(s),
                                        vp(stream)))
        self.K = k_end - k_begin
        return s.value

    def paint(self, d_out, stream=0):
        check(lib().relate_plan_paint(self._h, vp(d_out), vp(stream)))

    def out_bytes(self):
        return self.W * self.K * record_bytes(self.N)

    def repaint_layout(self, window):
        """(total_rows, pitch, row_offsets[K+1]) of the topology of one window (RePaintSection for all targets)."""
        rows, pitch = C.c_int64(), C.c_int()
        off = np.zeros(self.K + 1, dtype=np.int64)
        check(lib().relate_plan_repaint_layout(self._h, window, C.byref(rows), C.byref(pitch),
                                               off.ctypes.data_as(C.POINTER(C.c_int64))))
        return rows.value, pitch.value, off

    def repaint(self, window, d_stones, d_topology, d_logscales, stream=0):
        """FastPainting::RePaintSection (src/fast_painting.cpp:654-1126) of every prepared target for one window;
        d_stones = slab `window` of what paint() wrote (device pointers)."""
        check(lib().relate_plan_repaint(self._h, window, vp(d_stones), vp(d_topology), vp(d_logscales), vp(stream)))

    def distance_matrix(self, N, d_topology, pitch, d_logscales, total_rows, row_index, interpolate, weight_left,
                        weight_right, d_matrix, stream=0):
        """DistanceMeasure::GetMatrix (src/anc_builder.cpp:113-211) for one SNP; per-row arguments are host arrays."""
        ri = np.ascontiguousarray(row_index, dtype=np.int64)
        fl = np.ascontiguousarray(interpolate, dtype=np.uint8)
        wl = np.ascontiguousarray(weight_left, dtype=np.float64)
        wr = np.ascontiguousarray(weight_right, dtype=np.float64)
        check(lib().relate_plan_distance_matrix(self._h, N, vp(d_topology), pitch, vp(d_logscales), total_rows,
                                                ri.ctypes.data_as(i64p), fl.ctypes.data_as(u8p),
                                                wl.ctypes.data_as(f64p), wr.ctypes.data_as(f64p), vp(d_matrix),
                                                vp(stream)))

    def derived_counts(self):
        """nd_k of every haplotype of the loaded chunk (int32 array of N)."""
        nd = np.zeros(self.N, dtype=np.int32)
        check(lib().relate_plan_derived_counts(self._h, nd.ctypes.data_as(i32p)))
        return nd

    def paint_to_files(self, theta_painter, theta_data, k_begin, k_end, fds):
        """Paints [k_begin,k_end) and pwrite()s record k of window w at k*R of the open file descriptor fds[w]."""
        arr = (C.c_int * len(fds))(*fds)
        check(lib().relate_plan_paint_to_files(self._h, float(theta_painter), float(theta_data), k_begin, k_end, arr,
                                               len(fds)))

    def timings(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        check(lib().relate_plan_timings(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"pack_ms": a.value, "prepare_ms": b.value, "paint_ms": c.value}

    def geometry(self):
        g, m = C.c_int(), C.c_int()
        check(lib().relate_plan_geometry(self._h, C.byref(g), C.byref(m)))
        return {"waves_per_target": g.value, "states_per_lane": m.value}

    def repaint_geometry(self):
        g, m, b = C.c_int(), C.c_int(), C.c_int()
        check(lib().relate_plan_repaint_geometry(self._h, C.byref(g), C.byref(m), C.byref(b)))
        return {"waves_per_target": g.value, "states_per_lane": m.value, "checkpoint_rows": b.value}


def distance_matrix_noalloc(device, N, d_topology, pitch, d_logscales, total_rows, row_index, interpolate, weight_left,
                            weight_right, d_matrix, stream=0):
    """relate_distance_matrix_noalloc: Plan.distance_matrix without a plan and without any allocation in the
    library (N <= 36864).  The four per-row arrays must be contiguous numpy arrays of the right types already
    (int64, uint8, float64, float64): nothing is converted or copied on the way in."""
    for a, dt in ((row_index, np.int64), (interpolate, np.uint8), (weight_left, np.float64), (weight_right, np.float64)):
        if a is not None and (a.dtype != dt or not a.flags.c_contiguous):
            raise TypeError("per-row arguments must be contiguous arrays of int64 / uint8 / float64 / float64")
    check(lib().relate_distance_matrix_noalloc(
        device, N, vp(d_topology), pitch, vp(d_logscales), total_rows, row_index.ctypes.data_as(i64p),
        interpolate.ctypes.data_as(u8p), None if weight_left is None else weight_left.ctypes.data_as(f64p),
        None if weight_right is None else weight_right.ctypes.data_as(f64p), vp(d_matrix), vp(stream)))


def fast_log(x, device=0):
    """GPU fast_log (src/fast_log.hpp:6-21) of a float32 array."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    check(lib().relate_fast_log(x.ctypes.data_as(f32p), y.ctypes.data_as(f32p), x.size, device))
    return y
