"""ctypes binding of libmpbp_hip.so (C ABI: include/mpbp_hip.h).  There is NO CPU fallback: if the
HIP library is missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libmpbp_hip.so")
SOURCES = ["mpbp_hip.hip", "v2_engine.hip"]


def headers():
    """every header a translation unit can include: all of csrc/*.h plus the public C ABI header (a stale-object check that
    lists headers by hand silently misses new ones - round-3 review item 6)"""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "mpbp_hip.h")]


MPBP_TRUNC_THRESH, MPBP_TRUNC_BOND, MPBP_TRUNC_BOND_MAX, MPBP_TRUNC_BOND_THRESH = 0, 1, 2, 3


class MPBPError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmpbp_hip error {code}: {msg}")
        self.code = code


class Trunc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("mprime", C.c_int32), ("eps", C.c_double)]


class Desc(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_edges", C.c_int32), ("T", C.c_int32), ("q", C.c_int32),
                ("nbr_ptr", C.POINTER(C.c_int32)), ("in_edge", C.POINTER(C.c_int32)),
                ("out_edge", C.POINTER(C.c_int32)), ("max_bond", C.c_int32), ("device", C.c_int32),
                ("slot_of_edge", C.POINTER(C.c_int32)), ("n_slots", C.c_int32),
                ("ext_cores", C.c_void_p), ("ext_bonds", C.c_void_p), ("stream", C.c_void_p), ("periodic", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [("core_slot_doubles", C.c_int64), ("core_stride", C.c_int64), ("bonds_per_slot", C.c_int32),
                ("n_slots", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("maxerr", C.c_double), ("n_compress", C.c_int64), ("nan_flag", C.c_int32),
                ("capacity_flag", C.c_int32), ("jacobi_not_converged", C.c_int32), ("ms_total", C.c_float),
                ("ms_orth", C.c_float), ("n_orth_launches", C.c_int32), ("jacobi_sweeps", C.c_int64),
                ("jacobi_calls", C.c_int64)]


EXPORTS = ["mpbp_create", "mpbp_destroy", "mpbp_last_error", "mpbp_slab_layout", "mpbp_slab_pointers",
           "mpbp_set_factor", "mpbp_set_generic_factor", "mpbp_set_node_states", "mpbp_set_phi", "mpbp_set_psi", "mpbp_set_messages", "mpbp_get_bonds",
           "mpbp_get_messages", "mpbp_reset_messages", "mpbp_sweep", "mpbp_beliefs", "mpbp_get_belief_train", "mpbp_pair_beliefs",
           "mpbp_free_energy", "mpbp_logz", "mpbp_allgather_slots", "mpbp_twovar_marginals", "mpbp_set_profiling", "mpbp_phase_profile", "mpbp_selftest_gemm", "mpbp_selftest_qr", "mpbp_selftest_qr_bench",
           "mpbp_selftest_jacobi_bench", "mpbp_selftest_svd", "mpbp_selftest_qr_batched", "mpbp_selftest_qr_batched_seq", "mpbp_selftest_jacobi_grid", "mpbp_selftest_jacobi_block"]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU): one object per source
    (compiled concurrently, only when stale), then one link."""
    hdrs = headers()
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value", "-fPIC"]
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        deps = [src] + hdrs
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in deps):
            procs.append((s, subprocess.Popen(["hipcc"] + flags + ["-c", "-o", obj, src], stdout=subprocess.PIPE,
                                              stderr=subprocess.STDOUT, text=True)))
    for s, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n" + out)
        if verbose:
            print(out)
    if procs or not os.path.exists(LIB_PATH) or any(os.path.getmtime(LIB_PATH) < os.path.getmtime(o) for o in objs):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_lib = None


def lib():
    """Load the library (compiling it first if the binary is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # a fresh checkout: compile the library once (this is the product itself, not a fallback)
        try:
            build()
        except Exception as e:           # no hipcc, or the compile failed
            raise MPBPError(-100, f"{LIB_PATH} not found and could not be built ({e}): run "
                                  "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).  "
                                  "There is no CPU fallback.") from e
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise MPBPError(-101, f"symbol {name} missing from {LIB_PATH}")
    L.mpbp_last_error.restype = C.c_char_p
    L.mpbp_last_error.argtypes = [C.c_void_p]
    L.mpbp_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Desc)]
    L.mpbp_destroy.argtypes = [C.c_void_p]
    L.mpbp_destroy.restype = None
    L.mpbp_slab_layout.argtypes = [C.c_void_p, C.POINTER(Layout)]
    L.mpbp_slab_pointers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.mpbp_set_factor.argtypes = [C.c_void_p, C.c_int32, C.c_int32, ip, C.c_int32, dp, dp, dp, dp]
    L.mpbp_set_generic_factor.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, dp]
    L.mpbp_set_node_states.argtypes = [C.c_void_p, ip]
    L.mpbp_set_phi.argtypes = [C.c_void_p, dp]
    L.mpbp_set_psi.argtypes = [C.c_void_p, dp]
    L.mpbp_set_messages.argtypes = [C.c_void_p, ip, lp, dp]
    L.mpbp_get_bonds.argtypes = [C.c_void_p, ip]
    L.mpbp_get_messages.argtypes = [C.c_void_p, lp, dp]
    L.mpbp_reset_messages.argtypes = [C.c_void_p]
    L.mpbp_sweep.argtypes = [C.c_void_p, ip, C.c_int32, Trunc, C.c_double, C.POINTER(Stats)]
    L.mpbp_beliefs.argtypes = [C.c_void_p, dp]
    L.mpbp_pair_beliefs.argtypes = [C.c_void_p, dp, dp]
    L.mpbp_get_belief_train.argtypes = [C.c_void_p, C.c_int32, ip, dp, C.c_int64]
    L.mpbp_free_energy.argtypes = [C.c_void_p, dp]
    L.mpbp_logz.argtypes = [C.c_void_p, dp, dp]
    L.mpbp_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    L.mpbp_twovar_marginals.argtypes = [C.c_void_p, ip, C.c_int32, C.c_int32, dp]
    L.mpbp_allgather_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.mpbp_phase_profile.argtypes = [C.c_void_p, dp, C.c_int32, C.c_int32]
    L.mpbp_selftest_gemm.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, dp, dp]
    L.mpbp_selftest_qr.argtypes = [C.c_int32, C.c_int32, C.c_int32, dp, dp]
    L.mpbp_selftest_qr_bench.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp]
    L.mpbp_selftest_svd.argtypes = [C.c_int32, C.c_int32, C.c_int32, dp, dp, dp]
    L.mpbp_selftest_jacobi_grid.argtypes = [C.c_int32, C.c_int32, C.c_int32, dp, dp, C.c_int32, ip]
    L.mpbp_selftest_jacobi_block.argtypes = [C.c_int32, C.c_int32, C.c_int32, dp, dp, C.c_int32, ip]
    L.mpbp_selftest_qr_batched.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, dp, dp]
    L.mpbp_selftest_qr_batched_seq.argtypes = [C.c_int32, C.c_int32, ip, C.c_int32, dp, dp, ip]
    _lib = L
    return L


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().mpbp_last_error(ctx)
        raise MPBPError(rc, msg.decode() if msg else "?")
