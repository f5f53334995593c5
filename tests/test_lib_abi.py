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


def test_ctypes_structs_match_the_c_header(tmp_path):
    """Sizes and key field offsets of the ABI structs as a C compiler sees include/mpbp_hip.h against the ctypes mirror
    (a drifted struct would silently shift `periodic` / `stream` / the stats fields)."""
    import ctypes as C
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no C compiler")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mpbp_hip.h"\nint main(void) {\n'
                   '  printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mpbp_desc), sizeof(mpbp_trunc), sizeof(mpbp_stats),\n'
                   '         sizeof(mpbp_layout), offsetof(mpbp_desc, periodic), offsetof(mpbp_desc, stream),\n'
                   '         offsetof(mpbp_stats, jacobi_calls));\n  return 0;\n}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    L = mpbp_amd._lib
    want = [C.sizeof(L.Desc), C.sizeof(L.Trunc), C.sizeof(L.Stats), C.sizeof(L.Layout), L.Desc.periodic.offset,
            L.Desc.stream.offset, L.Stats.jacobi_calls.offset]
    assert got == want


def test_every_included_header_is_in_the_staleness_check():
    """Round-3 review item 6: `build()` recompiles an object when a header is newer - so every `#include "..."` reachable
    from the two translation units must be in `_lib.headers()` (csrc/cq_kernels.h was not, in round 3)."""
    L = mpbp_amd._lib
    known = {os.path.realpath(h) for h in L.headers()}
    assert all(os.path.exists(h) for h in known)
    seen, todo = set(), [os.path.join(L.CSRC, s) for s in L.SOURCES]
    while todo:
        f = todo.pop()
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), flags=re.M):
            cands = [os.path.join(os.path.dirname(f), inc), os.path.join(L.CSRC, inc), os.path.join(ROOT, "include", inc)]
            hit = [os.path.realpath(c) for c in cands if os.path.exists(c)]
            assert hit, f'{f} includes "{inc}", which is not in the tree'
            if hit[0] not in seen:
                seen.add(hit[0])
                todo.append(hit[0])
    assert seen, "no includes parsed"
    assert seen <= known, f"headers missing from the staleness check: {sorted(seen - known)}"
