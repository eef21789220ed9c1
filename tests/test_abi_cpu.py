"""CPU: the C-ABI library loads and exports every symbol include/tnr_hip.h declares (no compute calls)."""
import ctypes
import os
import re

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
