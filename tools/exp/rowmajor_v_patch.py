"""EXP builds only (make -C omg_amd/csrc EXP=1; tools/exp/attn_v7.h): run the PRODUCT's Python unchanged with the self-attention calls going
to attn_fwd_kernel7 — V read row-major out of the fused QKV projection, no omg_transpose_v — so that round 5 can time the whole benchmark
with it before the operand gets its place in the C ABI:

    python tools/exp/run_patched.py bench.py --steps 2 --warmup 1 --dedup-steps 0 --no-cpu-baseline

How: ``ops.transpose_v`` hands back a stand-in (no kernel runs) for the calls that produce the MFMA key order without a caller-owned output
buffer — the self-attention's V (more than 128 keys; the cached text / image-prompt projections keep their V^T); ``ops.attention`` sees the
stand-in, passes V to the library through the experiment's side door (omg_debug_set_attn_v) and forces variant 7 for that one launch.
Nothing here is imported by omg_amd, bench.py or the tests' product paths."""
import ctypes

import torch

from omg_amd import _lib as L
from omg_amd import ops


VARIANT = int(__import__("os").environ.get("OMG_ATTN_EXP_VARIANT", "7"))      # 8 = with the three-address asm first MFMA (tools/exp/attn_v7.h)


class RowMajorV:
    """What ``ops.attention`` needs of a V^T tensor (a non-null pointer, the padded key count) around the row-major V view."""

    def __init__(self, v: torch.Tensor, heads: int):
        self.v = v
        self.shape = (v.shape[0], heads, 64, (v.shape[1] + 63) // 64 * 64)

    def data_ptr(self) -> int:
        return self.v.data_ptr()


def install() -> None:
    lib = L.lib()
    if not hasattr(lib, "omg_debug_set_attn_v"):
        raise SystemExit("tools/exp/rowmajor_v_patch.py needs an EXP build of libomg_hip.so (make -C omg_amd/csrc EXP=1 DEV=1)")
    lib.omg_debug_set_attn_v.argtypes, lib.omg_debug_set_attn_v.restype = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64], None
    transpose_v, attention = ops.transpose_v, ops.attention

    def transpose_v_(v, heads, nkv_pad=None, out=None, mfma_order=True):
        if mfma_order and out is None and nkv_pad is None and v.shape[1] > 128 and v.stride(2) == 1 and v.stride(1) % 8 == 0:
            return RowMajorV(v, heads)
        return transpose_v(v, heads, nkv_pad, out, mfma_order)

    def attention_(q, k, vt, heads, scale, **kw):
        if not isinstance(vt, RowMajorV):
            return attention(q, k, vt, heads, scale, **kw)
        v = vt.v
        lib.omg_debug_set_attn_v(v.data_ptr(), v.stride(1), v.stride(0))
        lib.omg_debug_set_attn_variant(VARIANT)
        try:
            return attention(q, k, vt, heads, scale, **kw)
        finally:
            lib.omg_debug_set_attn_variant(0)

    ops.transpose_v, ops.attention = transpose_v_, attention_
