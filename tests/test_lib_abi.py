"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/mpbp_hip.h declares
(no compute calls - there is no GPU on the CPU runner)."""
import os
import re

import mpbp_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols():
    path = mpbp_amd.build()
    assert os.path.exists(path)
    lib = mpbp_amd._lib.lib()
    header = open(os.path.join(ROOT, "include", "mpbp_hip.h")).read()
    declared = set(re.findall(r"\b(mpbp_[a-z_0-9]+)\s*\(", header))
    declared -= {"mpbp_ctx"}
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mpbp_hip.h but not exported"
    assert declared == set(mpbp_amd._lib.EXPORTS)


def test_create_reports_errors_without_aborting():
    import ctypes as C
    lib = mpbp_amd._lib.lib()
    h = C.c_void_p()
    d = mpbp_amd._lib.Desc()      # all zero: invalid
    rc = lib.mpbp_create(C.byref(h), C.byref(d))
    assert rc == -1
    assert b"invalid descriptor" in lib.mpbp_last_error(None)
