"""The C-ABI library loads and exports every symbol include/orbhip.h declares (no compute calls, no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "orbhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:orbx|orbm|orbf|lba|liba|orb|pose|bow)_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_entry_points():
    names = _declared()
    assert "orbx_create" in names and "orbx_extract" in names and "orbx_extract_batch_dev" in names


def test_product_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    so = os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd", "liborbhip.so")
    if not os.path.exists(so):
        ge.build()
    lib = ctypes.CDLL(so)  # loads without a GPU (libamdhip64 is in the image)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_hint_entry_points_reject_unknown_bits():
    """lba_build_system_hint / pose_optimize_hint validate the hint word before anything else (argument check only: no GPU work)."""
    lib = ctypes.CDLL(os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd", "liborbhip.so"))
    lib.lba_build_system_hint.restype = ctypes.c_int
    lib.lba_build_system_hint.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
    assert lib.lba_build_system_hint(None, 1, None, 8, None) == -3          # ORB_E_INVALID: unknown hint bit
    assert lib.lba_build_system_hint(None, 1, None, 1, None) == -3          # known hint, null problem: the ordinary argument check
    lib.pose_optimize_hint.restype = ctypes.c_int
    lib.pose_optimize_hint.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_uint, ctypes.c_void_p]
    assert lib.pose_optimize_hint(None, None, None, 1, 1, None, 1, None, None, None, 1, None) == -3   # LBA_HINT_MONO_PINHOLE is not a pose_optimize hint


def test_loader_has_no_cpu_fallback(tmp_path, monkeypatch):
    from orbhip import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.OrbHipError):
        _lib.load()
