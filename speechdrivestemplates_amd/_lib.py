"""ctypes binding of libsdt_hip.so (the C ABI declared in include/sdt_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent, loading
raises; if a kernel call returns an error status, ``check`` raises RuntimeError with the library's
message (the reference signals errors with Python exceptions, core/networks/__init__.py:14-19).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDT_HIP_LIB: developer tools only (tools/conv_bench.py points it at the -DSDT_TUNING build); the package itself never sets it
LIB_PATH = os.environ.get("SDT_HIP_LIB") or os.path.join(_HERE, "lib", "libsdt_hip.so")
MAX_TAPS = 20
ABI_VERSION = 5  # sdt_abi_version() of the library this binding was written against


class ConvGeom(C.Structure):
    """Mirror of ``sdt_conv_geom`` (include/sdt_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in
                ("B", "Hi", "Wi", "Cin", "Ho", "Wo", "Hy", "Wy", "Cout", "sy", "sx", "osy", "osx", "ooy", "oox",
                 "ntaps", "Tw")] + [("dy", C.c_int32 * MAX_TAPS), ("dx", C.c_int32 * MAX_TAPS), ("wt", C.c_int32 * MAX_TAPS)]


class NormBwd(C.Structure):
    """Mirror of ``sdt_norm_bwd`` (include/sdt_hip.h)."""
    _fields_ = [("y", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("sums", C.c_void_p), ("slope", C.c_float), ("groups", C.c_int32)]


class WtDesc(C.Structure):
    """Mirror of ``sdt_wt_desc`` (include/sdt_hip.h)."""
    _fields_ = [("w", C.c_void_p), ("wt", C.c_void_p), ("w16", C.c_void_p), ("wt16", C.c_void_p), ("cout", C.c_int32), ("taps", C.c_int32),
                ("cin", C.c_int32), ("tile_begin", C.c_int32), ("planes", C.c_int32), ("reserved", C.c_int32)]


class ChainLayer(C.Structure):
    """Mirror of ``sdt_chain1d_layer`` (include/sdt_hip.h)."""
    _fields_ = [("Ti", C.c_int32), ("To", C.c_int32), ("Cin", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("in_mode", C.c_int32), ("src_a", C.c_int32), ("src_b", C.c_int32), ("reserved", C.c_int32),
                ("w", C.c_void_p), ("wt", C.c_void_p), ("y", C.c_void_p), ("x", C.c_void_p), ("dy", C.c_void_p), ("dx", C.c_void_p)]


CHAIN_PLAIN, CHAIN_NORM, CHAIN_UPADD = 0, 1, 2
_p, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_G = C.POINTER(ConvGeom)

# name -> argtypes; every function returns int status.  Keep in lock-step with include/sdt_hip.h
SIGNATURES = {
    "sdt_conv_taps_f32": [_p, _p, _p, _p, _G, _p],
    "sdt_conv_dw_f32": [_p, _p, _p, _G, _p],
    "sdt_conv_dw_det_f32": [_p, _p, _p, _G, _p, _i64, _p],
    "sdt_conv_taps_splitk_hint": [_G],
    "sdt_conv_taps_splitk_f32": [_p, _p, _p, _p, _G, _i, _p, _p],
    "sdt_splitk_reduce_f32": [_p, _p, _p, _i64, _i, _i, _p],
    "sdt_conv_taps_variant": [_G],
    "sdt_conv1d_small_used": [_G, _i, _i],
    "sdt_conv_dw_variant": [_G],
    "sdt_weight_transpose_f32": [_p, _p, _i, _i, _i, _p],
    "sdt_weight_transpose_batched_f32": [_p, _i, _i, _p],
    "sdt_set_conv_math": [_i],
    "sdt_col_sum_f32": [_p, _p, _i64, _i, _p],
    "sdt_colnorm_fwd_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _f, _f, _f, _i, _p],
    "sdt_colnorm_fwd_t": [_p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _f, _f, _f, _i, _p],
    "sdt_conv_taps_stats_supported": [_G, _i],
    "sdt_conv_taps_stats_f32": [_p, _p, _p, _p, _G, _p, _i, _p],
    "sdt_colnorm_eval_f32": [_p, _p, _p, _p, _p, _p, _i64, _i, _f, _f, _p],
    "sdt_colnorm_bwd_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _f, _i, _p],
    "sdt_colnorm_bwd_t": [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _f, _i, _p],
    "sdt_conv_taps_multi_f32": [_p, _p, _p, _G, _i, _i, _p, C.POINTER(NormBwd), _p],
    "sdt_l0_block_fwd_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _f, _p],
    "sdt_l0_block_bwd_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "sdt_l0_block_fwd_t": [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _f, _p],
    "sdt_l0_block_bwd_t": [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "sdt_rownorm_fwd_f32": [_p, _p, _p, _p, _i64, _i, _f, _f, _p],
    "sdt_rownorm_bwd_f32": [_p, _p, _p, _p, _p, _i64, _i, _f, _p],
    "sdt_rownorm_slabs_fwd_f32": [_p, _i, _p, _p, _p, _p, _i64, _i, _f, _f, _p],
    "sdt_resize_concat_fwd_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "sdt_resize_concat_bwd_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "sdt_upsample_add_fwd_f32": [_p, _p, _p, _i, _i, _i, _i, _p],
    "sdt_conv_dw_group_plan": [C.POINTER(_G), _i, _p, C.POINTER(C.c_int64)],
    "sdt_conv_dw_group_f32": [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i, _p, _p, _p, _p],
    "sdt_chain1d_supported": [C.POINTER(ChainLayer), _i, _i],
    "sdt_chain1d_fwd_f32": [C.POINTER(ChainLayer), _i, _p, _p, _i, _f, _f, _i, _p, _p, _p],
    "sdt_chain1d_bwd_f32": [C.POINTER(ChainLayer), _i, _p, _i, _f, _f, _i, _i, _p, _p, _p],
    "sdt_upsample_add_bwd_f32": [_p, _p, _i, _i, _i, _i, _p],
    "sdt_l1_loss_fwd_f32": [_p, _p, _i64, _f, _p, _p, _p],
    "sdt_l1_loss_bwd_f32": [_p, _p, _p, _i64, _f, _p, _p],
    "sdt_mse_const_fwd_f32": [_p, _i64, _f, _f, _p, _p],
    "sdt_mse_const_bwd_f32": [_p, _p, _i64, _f, _f, _p, _p],
    "sdt_code_kl_fwd_f32": [_p, _p, _i, _i, _i, _f, _p, _p, _p, _p],
    "sdt_code_kl_bwd_f32": [_p, _p, _p, _p, _i, _i, _i, _f, _p, _p],
    "sdt_final_metrics_f64": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p],
    "sdt_adam_step_f32": [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _f, _p, _p],
    "sdt_stft_frames_f32": [_p, _p, _i, _i, _i, _p],
    "sdt_mel_fb_f32": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "sdt_rows_scatter_add_f32": [_p, _p, _p, _i, _i, _i, _p],
    "sdt_time_diff_fwd_f32": [_p, _p, _i, _i, _i, _p],
    "sdt_time_diff_bwd_f32": [_p, _p, _i, _i, _i, _p],
    "sdt_clip_poses_prepare_f32": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "sdt_rows_gather_f32": [_p, _p, _p, _i, _i, _i64, _p],
    "sdt_convsk_supported": [_G, _i],
    "sdt_convsk_grid": [],
    "sdt_convsk_set_wg_per_cu": [_i],
    "sdt_convsk_set_f32_split": [_i],
    "sdt_convsk_set_reserved_slots": [_i],
    "sdt_convsk_plan_build": [_G, _i, _i, _i, _p, _i64],
    "sdt_convsk_dw_supported": [_G],
    "sdt_convsk_dw_plan_build": [_G, _p, _i64],
    "sdt_convsk_dw_f32": [_p, _p, _p, _p, _p, _p, _i64, _i64, _p],
    "sdt_convsk_f32": [_p, _p, _p, _p, _p, _p, _p, C.c_uint, _p, C.POINTER(NormBwd), _i64, _i64, _i64, _p],
    "sdt_convsk_bf16": [_p, _p, _p, _p, _p, _p, _p, C.c_uint, _p, C.POINTER(NormBwd), _i64, _i64, _i64, _p],
    "sdt_convsk_f32_w3": [_p, _p, _p, _p, _p, _p, _p, C.c_uint, _p, C.POINTER(NormBwd), _i64, _i64, _i64, _p],
    "sdt_convsk_supported_t": [_G, _i, _i],
    "sdt_convsk_plan_build_t": [_G, _i, _i, _i, _i, _i, _p, _i64],
    "sdt_convsk_dw_supported_t": [_G, _i],
    "sdt_convsk_dw_plan_build_t": [_G, _i, _p, _i64],
    "sdt_convsk_dw_bf16": [_p, _p, _p, _p, _p, _p, _i64, _i64, _p],
    "sdt_convsk_set_spin_limit": [C.c_uint],
    "sdt_convsk_set_k_order": [_i],
    "sdt_convsk_set_dw_wide_tiles": [_i],
}
F32, BF16 = 0, 1  # enum sdt_dtype

_lib = None


def load():
    """dlopen the kernel library and bind every exported entry point (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so: it must be in the process BEFORE our library is dlopen'ed, otherwise the
    # loader maps /opt/rocm's copy as well and the two HIP runtimes do not share devices / streams.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        # fresh checkout (the .so is git-ignored): compile it now if the ROCm toolchain is here (hipcc, ~10 s); this is a
        # build step, not a fallback -- without the library nothing in this package computes
        entry = os.path.join(os.path.dirname(_HERE), "__graft_entry__.py")
        if os.path.exists(entry) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) and not os.environ.get("SDT_NO_AUTOBUILD"):
            import importlib.util
            spec = importlib.util.spec_from_file_location("_sdt_graft_entry", entry)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            os.environ["SDT_NO_AUTOBUILD"] = "1"  # build() calls load() again
            try:
                mod.build()
            finally:
                os.environ.pop("SDT_NO_AUTOBUILD", None)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsdt_hip.so not found at %s -- run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first; "
            "this package has no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.sdt_abi_version.restype = C.c_int
    if lib.sdt_abi_version() != ABI_VERSION:  # checked BEFORE binding: a stale .so fails here, not with an AttributeError on a new symbol
        raise ImportError("libsdt_hip.so at %s has ABI version %d, this package needs %d -- rebuild it (python __graft_entry__.py)"
                          % (LIB_PATH, lib.sdt_abi_version(), ABI_VERSION))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.sdt_conv_dw_group_plan_bytes.argtypes = [_i]
    lib.sdt_conv_dw_group_plan_bytes.restype = C.c_int64
    lib.sdt_conv_dw_workspace_bytes.argtypes = [_G]
    lib.sdt_conv_dw_workspace_bytes.restype = C.c_int64
    lib.sdt_convsk_plan_bytes.argtypes = [_G, _i]
    lib.sdt_convsk_plan_bytes.restype = C.c_int64
    lib.sdt_convsk_plan_bytes_t.argtypes = [_G, _i, _i]
    lib.sdt_convsk_plan_bytes_t.restype = C.c_int64
    lib.sdt_convsk_dw_plan_bytes_t.argtypes = [_G, _i]
    lib.sdt_convsk_dw_plan_bytes_t.restype = C.c_int64
    lib.sdt_convsk_get_spin_limit.argtypes = []
    lib.sdt_convsk_get_spin_limit.restype = C.c_uint
    lib.sdt_convsk_dw_plan_bytes.argtypes = [_G]
    lib.sdt_convsk_dw_plan_bytes.restype = C.c_int64
    lib.sdt_convsk_dw_workspace_bytes.argtypes = []
    lib.sdt_convsk_dw_workspace_bytes.restype = C.c_int64
    lib.sdt_convsk_workspace_bytes.argtypes = []
    lib.sdt_convsk_workspace_bytes.restype = C.c_int64
    lib.sdt_last_error.restype = C.c_char_p
    lib.sdt_get_conv_math.restype = C.c_int
    _lib = lib
    return lib


def has_tuning():
    """True when the loaded library is the -DSDT_TUNING build (timeline stamps, fault injection, ablation instantiations)."""
    return hasattr(load(), "sdt_debug_convsk_mute_range")


def check(status):
    if status != 0:
        raise RuntimeError("libsdt_hip: %s (status %d)" % (load().sdt_last_error().decode(), status))
