"""ctypes binding of ``libomg_hip.so`` (C ABI declared in ``include/omg_hip.h``).

There is deliberately NO fallback: if the shared library has not been built
(``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C omg_amd/csrc``)
importing :func:`lib` raises — the product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OMG_HIP_LIB") or os.path.join(_HERE, "csrc", "libomg_hip.so")   # override: A/B builds in tools/

OMG_F16, OMG_BF16, OMG_F32 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GEGLU = 0, 1, 2
MAX_CONCEPTS = 8

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32),
        ("A", c_vp), ("lda", c_i64), ("W", c_vp), ("ldw", c_i64),
        ("A2", c_vp), ("lda2", c_i64), ("W2", c_vp), ("ldw2", c_i64),
        ("K2", c_i32), ("a2_col_block", c_i32),
        ("groups", c_i32), ("rows_per_group", c_i32),
        ("group_adapter", c_vp), ("w_adapter_stride", c_i64), ("w2_adapter_stride", c_i64),
        ("bias", c_vp), ("group_bias", c_vp), ("ldgb", c_i64),
        ("residual", c_vp), ("ldr", c_i64),
        ("act", c_i32), ("out_scale", c_f32),
        ("C", c_vp), ("ldc", c_i64),
    ]


class GemmMx8Args(C.Structure):
    _fields_ = [
        ("dtype", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32),
        ("A", c_vp), ("lda", c_i64), ("a_scale", c_vp), ("sa_ld", c_i32),
        ("W", c_vp), ("ldw", c_i64), ("w_scale", c_vp), ("sw_ld", c_i32),
        ("groups", c_i32), ("rows_per_group", c_i32), ("group_adapter", c_vp),
        ("w_adapter_stride", c_i64), ("sw_adapter_stride", c_i64),
        ("bias", c_vp), ("residual", c_vp), ("ldr", c_i64),
        ("act", c_i32), ("out_scale", c_f32),
        ("C", c_vp), ("ldc", c_i64),
        ("c_scale", c_vp), ("sc_ld", c_i32),
    ]


class Conv2dMx8Args(C.Structure):
    _fields_ = [
        ("dtype", c_i32), ("B", c_i32), ("H", c_i32), ("W", c_i32), ("Cin", c_i32), ("Cout", c_i32),
        ("X", c_vp), ("x_scale", c_vp), ("Wq", c_vp), ("w_scale", c_vp),
        ("sw_ld", c_i32), ("act", c_i32),
        ("bias", c_vp), ("group_bias", c_vp), ("ldgb", c_i64), ("residual", c_vp),
        ("out_scale", c_f32), ("Y", c_vp),
    ]


class Conv2dArgs(C.Structure):
    _fields_ = [
        ("dtype", c_i32), ("B", c_i32), ("Hin", c_i32), ("Win", c_i32),
        ("C1", c_i32), ("C2", c_i32), ("Hout", c_i32), ("Wout", c_i32), ("Cout", c_i32),
        ("ksize", c_i32), ("stride", c_i32), ("upsample", c_i32),
        ("X1", c_vp), ("X2", c_vp), ("W", c_vp), ("bias", c_vp),
        ("group_bias", c_vp), ("ldgb", c_i64), ("residual", c_vp),
        ("out_scale", c_f32), ("Y", c_vp), ("act", c_i32),
    ]


class Conv2dF32Args(C.Structure):
    _fields_ = [
        ("B", c_i32), ("Hin", c_i32), ("Win", c_i32), ("Cin", c_i32),
        ("Hout", c_i32), ("Wout", c_i32), ("Cout", c_i32), ("ksize", c_i32), ("upsample", c_i32),
        ("X", c_vp), ("W", c_vp), ("bias", c_vp), ("residual", c_vp), ("Y", c_vp),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("dtype", c_i32), ("B", c_i32), ("heads", c_i32), ("Nq", c_i32), ("Nkv", c_i32),
        ("Q", c_vp), ("ldq", c_i64), ("q_bstride", c_i64),
        ("K", c_vp), ("ldk", c_i64), ("k_bstride", c_i64),
        ("Vt", c_vp), ("Nkv_pad", c_i32),
        ("qk_src", c_vp), ("scale", c_f32), ("accumulate", c_i32), ("out_scale", c_f32),
        ("O", c_vp), ("ldo", c_i64), ("o_bstride", c_i64),
        ("V", c_vp), ("ldv", c_i64), ("v_bstride", c_i64),       # ABI 6: row-major V of the self-attention (no transpose pass)
    ]


class StepArgs(C.Structure):
    _fields_ = [
        ("C", c_i32), ("H", c_i32), ("W", c_i32), ("Hm", c_i32), ("Wm", c_i32),
        ("n_concepts", c_i32), ("fuse", c_i32), ("guidance_scale", c_f32),
        ("noise_pred", c_vp), ("region_pred", c_vp * MAX_CONCEPTS), ("masks", c_vp * MAX_CONCEPTS),
        ("coef", c_vp), ("step_idx", c_vp), ("advance", c_i32),
        ("latents", c_vp), ("out_dtype", c_i32), ("model_input_next", c_vp), ("fused_noise_out", c_vp),
    ]


# every symbol include/omg_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "omg_abi_version": (c_i32, []),
    "omg_last_error": (C.c_char_p, []),
    "omg_gemm": (c_i32, [C.POINTER(GemmArgs), c_vp]),
    "omg_quant_mx8": (c_i32, [c_i32, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_i32, c_vp]),
    "omg_gemm_mx8": (c_i32, [C.POINTER(GemmMx8Args), c_vp]),
    "omg_conv2d_mx8": (c_i32, [C.POINTER(Conv2dMx8Args), c_vp]),
    "omg_groupnorm_mx8": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "omg_conv2d": (c_i32, [C.POINTER(Conv2dArgs), c_vp]),
    "omg_conv2d_f32": (c_i32, [C.POINTER(Conv2dF32Args), c_vp]),
    "omg_cast_f32": (c_i32, [c_i32, c_vp, c_vp, c_i64, c_vp]),
    "omg_attn_fwd": (c_i32, [C.POINTER(AttnArgs), c_vp]),
    "omg_transpose_v": (c_i32, [c_i32, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "omg_groupnorm_ws_floats": (c_i64, [c_i32, c_i32, c_i32]),
    "omg_groupnorm": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "omg_layernorm": (c_i32, [c_i32, c_vp, c_i64, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "omg_layernorm_mx8": (c_i32, [c_i32, c_vp, c_i64, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_vp]),
    "omg_conv_in": (c_i32, [c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "omg_conv_out": (c_i32, [c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "omg_timestep_embedding": (c_i32, [c_i32, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "omg_silu": (c_i32, [c_i32, c_vp, c_vp, c_i64, c_vp]),
    "omg_softmax_rows": (c_i32, [c_i32, c_vp, c_i64, c_i64, c_i64, C.c_float, c_vp]),
    "omg_channel_mix": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp]),
    "omg_add_inplace": (c_i32, [c_i32, c_vp, c_vp, c_i64, c_vp]),
    "omg_copy2d": (c_i32, [c_i32, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "omg_fuse_cfg_step": (c_i32, [C.POINTER(StepArgs), c_vp]),
    "omg_scale_model_input": (c_i32, [c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "omg_gather_step": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "omg_attn_probs": (c_i32, [C.POINTER(AttnArgs), c_vp, c_vp]),
    "omg_attn_apply_probs": (c_i32, [c_i32, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp]),
    "omg_dwconv2d": (c_i32, [c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "omg_relu_linear_att_ws_floats": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "omg_relu_linear_att": (c_i32, [c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_i64, c_vp]),
    "omg_debug_set_glds": (None, [c_i32]),
    "omg_debug_set_gemm_variant": (None, [c_i32]),
    "omg_debug_choose_variant": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "omg_debug_set_mx8_split": (None, [c_i32]),
    "omg_debug_set_attn_variant": (None, [c_i32]),
}

_lib = None


class OmgHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OmgHipError(
            f"{LIB_PATH} not found: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "omg_amd has no CPU fallback."
        )
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): it must be in the process
    # BEFORE this library is, so that both resolve to ONE HIP runtime (streams, allocations and
    # graph capture are shared).  Loading in the other order yields hipErrorNoDevice at first launch.
    import torch  # noqa: F401

    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(l, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if l.omg_abi_version() != 6:
        raise OmgHipError("libomg_hip.so ABI version mismatch; rebuild")
    v = os.environ.get("OMG_GEMM_VARIANT")      # debugging/benchmarking aid: force one GEMM tile configuration
    if v:
        l.omg_debug_set_gemm_variant(int(v))
    v = os.environ.get("OMG_ATTN_VARIANT")      # ditto for the attention kernels (bit 16: attn_fwd_kernel7 without its XCD-aware block order)
    if v:
        l.omg_debug_set_attn_variant(int(v, 0))
    v = os.environ.get("OMG_MX8_SPLIT")         # ditto for the MX-fp8 GEMM: DMA split | debug bits << 8 (128 << 8: the one-tile-per-block form of the Linear kernel)
    if v:
        l.omg_debug_set_mx8_split(int(v, 0))
    _lib = l
    return l


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().omg_last_error()
        raise OmgHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
