"""ctypes binding of librelate_gpu.so (include/relate_gpu.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RELATE_GPU_LIB points experiments at another build of the same library (e.g. a timing-only ablation)
_PATH = os.environ.get("RELATE_GPU_LIB") or os.path.join(_HERE, "csrc", "librelate_gpu.so")

u8p = C.POINTER(C.c_uint8)
f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
vp = C.c_void_p


class RelateError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librelate_gpu error {code}: {msg}")
        self.code = code


def library_path():
    return _PATH


_lib = None

# name -> (restype, argtypes); exactly the functions include/relate_gpu.h declares
PROTOTYPES = {
    "relate_last_error": (C.c_char_p, []),
    "relate_device_count": (C.c_int, []),
    "relate_record_bytes": (C.c_size_t, [C.c_int]),
    "relate_paint_chunk": (C.c_int, [C.c_char_p, C.c_int, f64p, f64p, C.c_int]),
    "relate_paint_stones": (C.c_int, [C.c_int, C.c_int, u8p, f64p, C.c_double, C.c_double, C.c_int, i32p, C.c_int,
                                      C.c_int, u8p, C.c_int]),
    "relate_data_new": (C.c_int, [C.c_char_p] * 6 + [C.c_int, C.c_double, C.POINTER(vp)]),
    "relate_data_set_painting": (C.c_int, [vp, C.c_double, C.c_double]),
    "relate_data_dims": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "relate_data_free": (None, [vp]),
    "relate_painter_new": (C.c_int, [C.c_size_t, C.c_double, C.c_int, C.POINTER(vp)]),
    "relate_painter_paint_stepping_stones": (C.c_int, [vp, vp, C.c_char_p, C.c_size_t, i32p, C.c_int]),
    "relate_painter_free": (None, [vp]),
    "relate_plan_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "relate_plan_destroy": (None, [vp]),
    "relate_plan_configure": (C.c_int, [vp, C.c_int, C.c_int]),
    "relate_plan_load_device": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, i32p, vp]),
    "relate_plan_load_host": (C.c_int, [vp, C.c_int, C.c_int, u8p, f64p, C.c_int, i32p]),
    "relate_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "relate_plan_load_device_into": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, i32p, vp, C.c_size_t, vp]),
    "relate_plan_packed": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "relate_plan_load_packed": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_size_t, vp, C.c_int, i32p, vp]),
    "relate_plan_prepare": (C.c_int, [vp, C.c_double, C.c_double, C.c_int, C.c_int, i64p, vp]),
    "relate_plan_paint": (C.c_int, [vp, vp, vp]),
    "relate_plan_repaint_layout": (C.c_int, [vp, C.c_int, i64p, C.POINTER(C.c_int), i64p]),
    "relate_plan_repaint": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    "relate_plan_distance_matrix": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int64, i64p, u8p, f64p, f64p, vp, vp]),
    "relate_distance_matrix_noalloc": (C.c_int, [C.c_int, C.c_int, vp, C.c_int, vp, C.c_int64, i64p, u8p, f64p, f64p, vp, vp]),
    "relate_distance_new": (C.c_int, [vp, u8p, f64p, vp, C.c_int, C.POINTER(vp)]),
    "relate_distance_free": (None, [vp]),
    "relate_distance_get_topology_with_repaint": (C.c_int, [vp, C.c_int, vp]),
    "relate_distance_advance": (C.c_int, [vp, C.c_int]),
    "relate_distance_get_matrix": (C.c_int, [vp, C.c_int, vp, vp]),
    "relate_distance_state": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), i64p, f64p, f64p]),
    "relate_minmatch_threshold": (C.c_float, [C.c_double]),
    "relate_plan_minmatch_scan": (C.c_int, [vp, C.c_int, vp, C.c_float, f32p, C.c_int64, i32p, i32p, f32p, i64p, vp]),
    "relate_minmatch_apply_candidates": (C.c_int, [C.c_int, C.c_int64, i32p, i32p, f32p, f64p, C.c_double, vp, vp, vp, vp]),
    "relate_plan_derived_counts": (C.c_int, [vp, i32p]),
    "relate_shard_by_work": (C.c_int, [C.c_int, i32p, C.c_int, i32p]),
    "relate_plan_paint_to_files": (C.c_int, [vp, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "relate_release_host_staging": (None, []),
    "relate_plan_timings": (C.c_int, [vp, f32p, f32p, f32p]),
    "relate_plan_geometry": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "relate_plan_repaint_geometry": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "relate_fast_log": (C.c_int, [f32p, f32p, C.c_int, C.c_int]),
    "relate_synth_chunk": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_int, u8p, f64p]),
    "relate_write_chunk": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, u8p, f64p, C.c_int, i32p]),
    "relate_window_boundaries": (C.c_int, [C.c_int, C.c_int, u8p, C.c_double, C.c_int, i32p]),
    "relate_haps_open": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, C.POINTER(vp)]),
    "relate_haps_close": (None, [vp]),
    "relate_haps_dims": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "relate_haps_chunk_dims": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "relate_haps_chunk_windows": (C.c_int, [vp, C.c_int, i32p]),
    "relate_haps_positions": (C.c_int, [vp, i32p, i32p, f64p, f64p]),
    "relate_haps_rows": (C.c_int, [vp, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int)]),
    "relate_row_stride_words": (C.c_int, [C.c_int]),
    "relate_haps_write_chunks": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "relate_plan_load_haps_chunk": (C.c_int, [vp, vp, C.c_int, C.c_double]),
    "relate_plan_load_rows_host": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_uint64), f64p, C.c_double, C.c_int, i32p]),
}


def lib():
    """The loaded library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RelateError(-3, f"{_PATH} is missing: build it with `make -C relate_amd/csrc` "
                                  "(or __graft_entry__.build()); there is no CPU fallback")
        # One HIP runtime per process: torch ships its own libamdhip64.so.7; loading it first makes the
        # library bind to the same copy (same soname) instead of bringing in /opt/rocm's beside it, after
        # which torch would see no GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_PATH)
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RelateError(rc, lib().relate_last_error().decode(errors="replace"))


def record_bytes(N):
    return 56 + 8 * N
