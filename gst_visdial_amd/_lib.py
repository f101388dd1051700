"""ctypes binding of the C ABI declared in include/gstvd_hip.h (libgstvd_hip.so, gfx950).

There is NO fallback: if the shared library is missing or a call returns a non-zero status the
product raises.  (Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C gst_visdial_amd/csrc`.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgstvd_hip.so")
ABI_VERSION = 7

F32, BF16 = 0, 1
EPI_BIAS, EPI_ADD, EPI_GELU, EPI_DGELU, EPI_DROPOUT = 1, 2, 4, 8, 16
EPI_COLSUM, EPI_COLSUM_ACC = 32, 64
EPI_ADAMW = 128
LN_RESID, LN_EMBED, LN_IMAGE = 0, 1, 2

_vp, _i64, _i32, _f32, _u32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint32


class GemmDesc(C.Structure):
    _fields_ = [("A", _vp), ("B", _vp), ("C", _vp), ("bias", _vp), ("addend", _vp), ("aux", _vp),
                ("M", _i64), ("N", _i64), ("K", _i64),
                ("lda", _i64), ("ldb", _i64), ("ldc", _i64), ("ldadd", _i64), ("ldaux", _i64),
                ("batch", _i64), ("sA", _i64), ("sB", _i64), ("sC", _i64), ("sAdd", _i64), ("sAux", _i64),
                ("dtype_in", _i32), ("dtype_out", _i32), ("a_kmajor", _i32), ("b_kmajor", _i32), ("epilogue", _i32),
                ("alpha", _f32), ("dropout_p", _f32), ("site", _u32), ("rng", _vp)]


class LnDesc(C.Structure):
    _fields_ = [("mode", _i32), ("dtype", _i32), ("M", _i64), ("H", _i64),
                ("x", _vp), ("ldx", _i64), ("res", _vp), ("ldres", _i64),
                ("gamma", _vp), ("beta", _vp), ("eps", _f32),
                ("y", _vp), ("ldy", _i64), ("mean", _vp), ("rstd", _vp),
                ("p_pre", _f32), ("p_post", _f32), ("site_pre", _u32), ("site_post", _u32), ("rng", _vp),
                ("ids", _vp), ("segs", _vp), ("T", _i64), ("type_vocab", _i32),
                ("word", _vp), ("pos", _vp), ("tt", _vp), ("tt_ext", _vp),
                ("loc", _vp), ("w_loc", _vp), ("b_loc", _vp), ("pos_offset", _i64)]


class LnBwdDesc(C.Structure):
    _fields_ = [("f", LnDesc), ("dy", _vp), ("lddy", _i64), ("dres", _vp), ("lddres", _i64),
                ("dx", _vp), ("lddx", _i64), ("partial", _vp),
                ("dword", _vp), ("dpos", _vp), ("dtt", _vp), ("dtt_ext", _vp), ("nblk", _i64)]


class AttnDesc(C.Structure):
    _fields_ = [("Q", _vp), ("K", _vp), ("V", _vp), ("O", _vp), ("LSE", _vp), ("key_mask", _vp),
                ("ldq", _i64), ("ldk", _i64), ("ldv", _i64), ("ldo", _i64),
                ("B", _i32), ("nh", _i32), ("Lq", _i32), ("Lk", _i32), ("d", _i32), ("causal", _i32), ("dtype", _i32),
                ("mask_neg", _f32), ("scale", _f32), ("dropout_p", _f32), ("site", _u32), ("rng", _vp),
                ("dO", _vp), ("lddo", _i64), ("dQ", _vp), ("dK", _vp), ("dV", _vp),
                ("lddq", _i64), ("lddk", _i64), ("lddv", _i64), ("delta", _vp), ("kv_group", _i32),
                ("q_bstride", _i32), ("kv_bstride", _i32), ("drop_bits", _vp)]


class SampleDesc(C.Structure):
    _fields_ = [("logits", _vp), ("ld", _i64), ("dtype", _i32), ("B", _i32), ("V", _i32), ("top_k", _i32), ("temperature", _f32),
                ("u", _vp), ("out", _vp), ("out_stride", _i64), ("banned", _vp), ("banned_ld", _i64),
                ("hist", _vp), ("hist_ld", _i64), ("hist_T", _i32), ("ngram", _i32),
                ("ids_tm", _vp), ("ids_stride", _i64), ("cur_len", _i32), ("n_special", _i32), ("special", _i32 * 8),
                ("top_p", _f32), ("reserved_", _i32)]


class AdamFuse(C.Structure):
    """gstvd_adamw_fuse_t: the flat buffers and constants of the optimizer, for the weight-gradient launch that updates in its epilogue."""
    _fields_ = [("grad_base", _vp), ("param", _vp), ("m", _vp), ("v", _vp), ("shadow_bf16", _vp),
                ("step", _vp), ("beta1", _f32), ("beta2", _f32), ("eps", _f32), ("grad_scale", _f32), ("write_grad", _i32)]


class ColsumEntry(C.Structure):
    _fields_ = [("partial", _vp), ("out", _vp * 3), ("nblk", _i64), ("stride", _i64), ("H", _i64),
                ("nvec", _i32), ("accumulate", _i32 * 3), ("blk0", _i32)]


class SlabEntry(C.Structure):
    _fields_ = [("x", _vp), ("scratch", _vp), ("ldx", _i64), ("M", _i64), ("N", _i64), ("blk0", _i32), ("pad_", _i32)]


# name -> (restype, argtypes); every symbol include/gstvd_hip.h declares
SIGNATURES = {
    "gstvd_abi_version": (_i32, []),
    "gstvd_build_arch": (C.c_char_p, []),
    "gstvd_gemm": (_i32, [C.POINTER(GemmDesc), _vp]),
    "gstvd_gemm_splitk_ws_bytes": (_i64, [_i64, _i64, _i32]),
    "gstvd_gemm_splitk": (_i32, [_vp, _i32, _vp, _i64, _vp]),
    "gstvd_gemm_group_tile": (_i32, []),
    "gstvd_gemm_group_caps": (_i32, []),
    "gstvd_gemv_ln": (_i32, [C.POINTER(GemmDesc), _vp, _vp, _f32, _vp, _i64, _vp]),
    "gstvd_gemm_kernel_name": (_i32, [C.POINTER(GemmDesc), _i32, C.c_char_p, _i32]),
    "gstvd_gemm_grouped_kernel_name": (_i32, [_i32, _i32, _i32, _i32, C.c_char_p, _i32]),
    "gstvd_gemm_grouped": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "gstvd_ln_fwd": (_i32, [C.POINTER(LnDesc), _vp]),
    "gstvd_ln_bwd_blocks": (_i64, [_i64]),
    "gstvd_ln_bwd_blocks_for": (_i64, [_i64, _i64, _i32]),
    "gstvd_ln_bwd": (_i32, [C.POINTER(LnBwdDesc), _vp]),
    "gstvd_gemm_ln_fwd": (_i32, [C.POINTER(GemmDesc), C.POINTER(LnDesc), _vp]),
    "gstvd_gemm_ln_bwd": (_i32, [C.POINTER(GemmDesc), C.POINTER(LnBwdDesc), _vp]),
    "gstvd_gemm_ln_rows_per_block": (_i64, []),
    "gstvd_colsum_partials": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i32, _vp]),
    "gstvd_colsum_batched": (_i32, [_vp, _i64, _i64, _vp]),
    "gstvd_colsum_slabs_batched": (_i32, [_vp, _i64, _i64, _i32, _vp]),
    "gstvd_colsum_slabs": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp, _i64, _vp]),
    "gstvd_colsum": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp, _vp, _i64, _i32, _vp]),
    "gstvd_locgrad": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _i32, _vp]),
    "gstvd_attn_fwd": (_i32, [C.POINTER(AttnDesc), _vp]),
    "gstvd_attn_bwd": (_i32, [C.POINTER(AttnDesc), _vp]),
    "gstvd_ce_fwd": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "gstvd_ce_bwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _vp, _i64, _vp]),
    "gstvd_answer_scores": (_i32, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "gstvd_sample_topk": (_i32, [C.POINTER(SampleDesc), _vp]),
    "gstvd_vl_split": (_i32, [_vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _f32, _u32, _u32, _vp, _vp]),
    "gstvd_cast": (_i32, [_vp, _i32, _vp, _i32, _i64, _vp]),
    "gstvd_cast_ranges": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "gstvd_scale": (_i32, [_vp, _vp, _i64, _vp]),
    "gstvd_rng_advance": (_i32, [_vp, _vp]),
    "gstvd_dropout_mask": (_i32, [_vp, _i64, _f32, _u32, _vp, _vp]),
    "gstvd_adamw": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _f32, _f32, _f32, _vp, _f32, _i64, _vp]),
    "gstvd_adamw_bf16grad": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _f32, _f32, _f32, _vp, _f32, _i64, _vp]),
    "gstvd_adamw_blocks": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _f32, _f32, _f32, _vp, _f32, _i64, _vp, _i64, _vp, _vp]),
    "gstvd_gemm_grouped_adamw": (_i32, [_vp, _vp, _i64, _i64, C.POINTER(AdamFuse), _vp, _i64, _vp]),
    "gstvd_gemm_grouped_adamw_kernel_name": (_i32, [C.c_char_p, _i32]),
}

_STATUS = {-1: "GSTVD_E_DTYPE", -2: "GSTVD_E_SHAPE", -3: "GSTVD_E_ALIGN", -4: "GSTVD_E_NULL", -5: "GSTVD_E_UNSUPPORTED"}


class GstvdError(RuntimeError):
    pass


_lib = None


def load():
    """Load libgstvd_hip.so (once).  Raises if it is missing -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GstvdError(
            "gst_visdial_amd: HIP library %s not found; build it first (make -C gst_visdial_amd/csrc). "
            "There is no CPU fallback for the product path." % LIB_PATH)
    # torch first: its wheel bundles its own HIP runtime (libamdhip64).  If this library were dlopen'ed before torch, the
    # loader would pull in /opt/rocm's copy for it and torch would then bring a second runtime into the process -- kernels
    # registered with one, streams and allocations owned by the other (observed: hipErrorNoDevice on the first launch).
    # With torch loaded first the dependency resolves to the runtime already in the process.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.gstvd_abi_version() != ABI_VERSION:
        raise GstvdError("libgstvd_hip.so ABI %d != expected %d; rebuild" % (lib.gstvd_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


N_CALLS = [0]      # launching library calls made so far (each is >= 1 kernel launch): bench.py reports calls per decode token


def check(name, rc):
    N_CALLS[0] += 1
    if rc != 0:
        what = _STATUS.get(rc, "hipError_t %d" % rc if rc > 0 else "status %d" % rc)
        raise GstvdError("%s failed: %s" % (name, what))
