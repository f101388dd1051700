"""Diagnostic library for the probes in tools/ (NOT part of the product): builds gst_visdial_amd/lib/libgstvd_hip_diag.so
(`make -C gst_visdial_amd/csrc diag`, the same sources with -DGSTVD_DIAG) and points the ctypes binding at it for THIS process.
The diagnostic build adds what the product library must not contain: GSTVD_DIAG_ABLATE = 1 / 2 / 6 / 7 / 8 (timing-only
ablations of the 256-tile GEMM's K loop and epilogue -- wrong results by design), GSTVD_GEMM_ST=3 (in-kernel clock stamps) and
gstvd_debug_gemm_clock().  Usage: `import diag_lib; diag_lib.use()` before the first gst_visdial_amd op."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def use(build=True):
    from gst_visdial_amd import _lib
    path = os.path.join(ROOT, "gst_visdial_amd", "lib", "libgstvd_hip_diag.so")
    if build:       # always: make is incremental, and a library that merely EXISTS may predate the last source edit
        subprocess.run(["make", "-C", os.path.join(ROOT, "gst_visdial_amd", "csrc"), "-j4", "diag"], check=True)
    if _lib._lib is not None:
        raise RuntimeError("diag_lib.use() must run before the product library is loaded")
    _lib.LIB_PATH = path
    lib = _lib.load()
    lib.gstvd_debug_gemm_clock.restype = C.c_int32
    lib.gstvd_debug_gemm_clock.argtypes = [C.c_void_p, C.c_int32]
    return lib
