"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/gstvd_hip.h declares (no compute calls here -- there is no GPU in the CPU suite)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gstvd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef GSTVD_DIAG.*?#endif", "", src, flags=re.S)      # entry points of the diagnostic build (tools/ only)
    return sorted(set(re.findall(r"\b(gstvd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("gstvd_gemm", "gstvd_ln_fwd", "gstvd_ln_bwd", "gstvd_attn_fwd", "gstvd_attn_bwd", "gstvd_ce_fwd",
                 "gstvd_ce_bwd", "gstvd_adamw"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from gst_visdial_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = _lib.load()
    for s in declared_symbols():
        assert hasattr(lib, s), "libgstvd_hip.so does not export " + s
        assert s in _lib.SIGNATURES, "no ctypes signature for " + s
    assert set(_lib.SIGNATURES) == set(declared_symbols())
    assert lib.gstvd_abi_version() == _lib.ABI_VERSION
    assert lib.gstvd_build_arch() == b"gfx950"


def test_ops_refuse_cpu_tensors():
    import torch
    from gst_visdial_amd import ops, _lib
    a = torch.zeros(8, 8)
    with pytest.raises(_lib.GstvdError):
        ops.gemm(a, a, a, 8, 8, 8)


def test_product_library_carries_no_timing_ablations():
    """an environment variable must not be able to make the product compute wrong results.  The ablation
    selector and the in-kernel clock probe live in the -DGSTVD_DIAG build (lib/libgstvd_hip_diag.so, `make diag`, tools/ only)."""
    from gst_visdial_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    blob = open(_lib.LIB_PATH, "rb").read()
    for needle in (b"GSTVD_GEMM_ABLATE", b"GSTVD_DIAG_ABLATE", b"gstvd_debug_gemm_clock", b"g_clk256"):
        assert needle not in blob, needle
    assert b"GSTVD_GEMM_PC" in blob                                           # (the scan does see getenv strings)
    assert not hasattr(_lib.load(), "gstvd_debug_gemm_clock")
    assert "diag" not in os.path.basename(_lib.LIB_PATH)
