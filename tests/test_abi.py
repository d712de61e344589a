"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol include/relate_gpu.h
declares (and nothing is bound in Python that the header lacks), and fails loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import relate_amd as ra
from relate_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "relate_gpu.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(relate_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(ra.library_path())
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in relate_gpu.h but not exported"


def test_python_binding_matches_header():
    assert sorted(_lib.PROTOTYPES) == declared_functions()


def test_no_torch_or_cxx_types_in_header():
    text = open(HEADER).read()
    for bad in ("std::", "torch", "at::", "hipStream_t ", "template"):
        assert bad not in re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    # compiles as plain C
    subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", HEADER], check=True)


def test_record_bytes():
    for n in (2, 5, 2000, 5008):
        assert ra.lib().relate_record_bytes(n) == 56 + 8 * n == ra.record_bytes(n)


@pytest.mark.skipif(ra.lib().relate_device_count() > 0, reason="a GPU is present")
def test_compute_entry_points_fail_loudly_without_gpu(tmp_path):
    """No CPU fallback: every compute entry point reports RELATE_EGPU when no device is visible."""
    seq = np.full((10, 4), ord("0"), dtype=np.uint8)
    r = np.full(10, 0.01)
    wb = np.array([0, 10], dtype=np.int32)
    with pytest.raises(ra.RelateError) as e:
        ra.paint_stones(seq, r, 0.001, 0.001, wb)
    assert e.value.code == -3
    with pytest.raises(ra.RelateError) as e:
        ra.Plan(0)
    assert e.value.code == -3
    ra.write_chunk(tmp_path, 0, seq, r, wb)
    with pytest.raises(ra.RelateError) as e:
        ra.paint_chunk(tmp_path, 0)
    assert e.value.code == -3


def test_bad_arguments_are_rejected(tmp_path):
    with pytest.raises(ra.RelateError) as e:
        ra.paint_chunk(tmp_path / "nowhere", 0)
    assert e.value.code in (-2, -3)  # no such file (or no GPU, which is checked first)
    with pytest.raises(ra.RelateError):
        ra.FastPainting(1, 0.001)
    with pytest.raises(ra.RelateError):
        ra.FastPainting(10, 1.5)  # theta must be < 1 (src/fast_painting.hpp:23)
    with pytest.raises(ra.RelateError):
        ra.Data(tmp_path / "missing.hap", None, None, tmp_path / "missing.r", None, None)


def test_missing_library_is_an_error_not_a_fallback(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_PATH", "/nonexistent/librelate_gpu.so")
    with pytest.raises(ra.RelateError):
        _lib.lib()
