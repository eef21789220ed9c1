"""CPU: the C-ABI library loads and exports every symbol include/tnr_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

import tnr_hip as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "tnr_hip.h")).read()
    declared = set(re.findall(r"\b(tnr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(T.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libtnr_hip.so does not export %s" % name
    assert declared == set(T.EXPORTS), declared ^ set(T.EXPORTS)


def test_version_without_gpu():
    assert T.query("tnr_version") == 1


def test_missing_library_fails_loudly(monkeypatch):
    """No CPU / eager fallback anywhere on the product path: without libtnr_hip.so the first engine call raises."""
    import engine as E
    import tnr_hip as T
    monkeypatch.setattr(T, "_lib", None)
    monkeypatch.setattr(T, "LIB_PATH", "/nonexistent/libtnr_hip.so")
    with pytest.raises(T.TnrError, match="not built"):
        E.Engine(E.EngineConfig(n_layers=1, trainable_layers=(0,), num_teachers=1), device="cpu", max_batch=1)
