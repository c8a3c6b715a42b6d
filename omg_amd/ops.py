"""Thin torch-tensor wrappers over the C ABI (``include/omg_hip.h``).

PyTorch here is plumbing only: it owns device memory (``torch.empty``), the current
HIP stream and nothing else — every arithmetic op below is a hand-written gfx950
kernel in ``omg_amd/csrc``.  All wrappers launch on ``torch.cuda.current_stream()``
so they are captured by ``torch.cuda.graph`` (hipGraph) like any torch op.
Tensors must live on a ROCm device; a CPU tensor raises (no fallback).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib as L

_DT = {torch.float16: L.OMG_F16, torch.bfloat16: L.OMG_BF16}


def _dt(t: torch.Tensor, allow_f32: bool = False) -> int:
    if allow_f32 and t.dtype == torch.float32:
        return L.OMG_F32
    try:
        return _DT[t.dtype]
    except KeyError:
        raise L.OmgHipError(f"unsupported dtype {t.dtype}; use float16 or bfloat16") from None


def _dev(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise L.OmgHipError("omg_amd ops need tensors on the MI355X (cuda/hip device); there is no CPU fallback")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class KernelProfiler:
    """Opt-in per-launch timing with HIP events on the launch stream (used by bench.py's roofline leg).
    Off by default: the product path records nothing."""

    def __init__(self):
        self.records = []   # (kind, flops, start_event, end_event)

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def end(self, kind: str, flops: float, start, tag=None) -> None:
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        self.records.append((kind, flops, start, e, tag))

    def summary(self):
        """kind -> dict(launches, ms, flops); call after torch.cuda.synchronize()."""
        out = {}
        for kind, fl, s, e, _ in self.records:
            d = out.setdefault(kind, dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
        return out

    def by_tag(self):
        """(kind, tag) -> dict(launches, ms, flops), sorted by total time."""
        out = {}
        for kind, fl, s, e, tag in self.records:
            d = out.setdefault((kind, tag), dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
        return sorted(out.items(), key=lambda kv: -kv[1]["ms"])


_PROF: Optional[KernelProfiler] = None


def set_profiler(p: Optional[KernelProfiler]) -> None:
    global _PROF
    _PROF = p


class LoraSpec:
    """Second K-segment of a GEMM: ``C += A2 @ W2[adapter]^T`` (PEFT ``s * B(A(x))``)."""

    __slots__ = ("a2", "w2", "group_adapter", "a2_col_block")

    def __init__(self, a2: torch.Tensor, w2: torch.Tensor, group_adapter: Optional[torch.Tensor] = None,
                 a2_col_block: int = 0):
        self.a2, self.w2, self.group_adapter, self.a2_col_block = a2, w2, group_adapter, a2_col_block


def gemm(a: torch.Tensor, w: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, act: int = L.ACT_NONE, out: Optional[torch.Tensor] = None,
         out_scale: float = 1.0, group_bias: Optional[torch.Tensor] = None, groups: int = 1,
         lora: Optional[LoraSpec] = None, w_group_adapter: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[M, N_out] = epi(a[M,K] @ w[N,K]^T)``; see ``omg_gemm`` in include/omg_hip.h.

    ``w`` may be 3-D ``[n_adapters, N, K]`` together with ``w_group_adapter`` (int32 device
    tensor, one adapter id per group) — the LoRA-down projection of a batch whose samples
    use different adapters.
    """
    _dev(a)
    M, K = a.shape
    assert a.stride(1) == 1
    if w.dim() == 3:
        N = w.shape[1]
        w_stride = w.stride(0)
        ldw = w.stride(1)
    else:
        N = w.shape[0]
        w_stride = 0
        ldw = w.stride(0)
    assert w.shape[-1] == K and w.stride(-1) == 1
    n_out = N // 2 if act == L.ACT_GEGLU else N
    if out is None:
        out = torch.empty((M, n_out), dtype=a.dtype, device=a.device)
    assert out.stride(1) == 1 and out.shape[0] == M and out.shape[1] == n_out
    args = L.GemmArgs()
    args.dtype = _dt(a)
    args.M, args.N, args.K = M, N, K
    args.A, args.lda = a.data_ptr(), a.stride(0)
    args.W, args.ldw = w.data_ptr(), ldw
    args.groups = groups
    args.rows_per_group = M // groups
    adapter = w_group_adapter
    if lora is not None:
        a2, w2 = lora.a2, lora.w2
        args.A2, args.lda2 = a2.data_ptr(), a2.stride(0)
        args.K2 = w2.shape[-1]
        args.W2 = w2.data_ptr()
        if w2.dim() == 3:
            args.ldw2, args.w2_adapter_stride = w2.stride(1), w2.stride(0)
        else:
            args.ldw2, args.w2_adapter_stride = w2.stride(0), 0
        args.a2_col_block = lora.a2_col_block
        if lora.group_adapter is not None:
            adapter = lora.group_adapter
    args.group_adapter = _p(adapter)
    args.w_adapter_stride = w_stride
    args.bias = _p(bias)
    if group_bias is not None:
        args.group_bias, args.ldgb = group_bias.data_ptr(), group_bias.stride(0)
    if residual is not None:
        assert residual.stride(1) == 1
        args.residual, args.ldr = residual.data_ptr(), residual.stride(0)
    args.act = act
    args.out_scale = out_scale
    args.C, args.ldc = out.data_ptr(), out.stride(0)
    if _PROF is not None:
        t0 = _PROF.begin()
        L.check(L.lib().omg_gemm(C.byref(args), _stream()), "omg_gemm")
        _PROF.end("gemm", 2.0 * M * N * (K + args.K2), t0, ("lin", M, N, K, args.K2, groups if adapter is not None else 1, act))
        return out
    L.check(L.lib().omg_gemm(C.byref(args), _stream()), "omg_gemm")
    return out


class Mx8Tensor:
    """An MX-fp8 quantised 2-D operand: ``q`` uint8 [rows, K] (OCP e4m3), ``scales`` int32 [K/128, s_ld] (four E8M0 bytes per
    (row, 128-wide stage), stage-major — the layout omg_gemm_mx8 stages with one LDS-DMA per tile and stage)."""

    __slots__ = ("q", "scales", "rows", "K", "shape")

    def __init__(self, q: torch.Tensor, scales: torch.Tensor, shape=None):
        self.q, self.scales, self.rows, self.K = q, scales, q.shape[0], q.shape[1]
        self.shape = tuple(shape) if shape is not None else (q.shape[0], q.shape[1])      # logical shape (..., K) of the activation

    @property
    def device(self):
        return self.q.device


def quant_mx8(x: torch.Tensor, out: Optional[Mx8Tensor] = None) -> Mx8Tensor:
    """fp16 / bf16 [..., K] (rows with one common stride, unit inner stride) -> :class:`Mx8Tensor` (omg_quant_mx8)."""
    _dev(x)
    shape = x.shape
    x = x.reshape(-1, shape[-1])
    assert x.stride(1) == 1
    M, K = x.shape
    if K % 128 != 0:
        raise L.OmgHipError("MX-fp8 operands need K % 128 == 0")
    if out is None:
        s_ld = (M + 3) // 4 * 4
        out = Mx8Tensor(torch.empty((M, K), dtype=torch.uint8, device=x.device),
                        torch.zeros((K // 128, s_ld), dtype=torch.int32, device=x.device), shape)
    assert out.q.shape == (M, K) and out.q.stride(1) == 1 and out.scales.shape[0] == K // 128
    L.check(L.lib().omg_quant_mx8(_dt(x), x.data_ptr(), x.stride(0), M, K, out.q.data_ptr(), out.q.stride(0),
                                  out.scales.data_ptr(), out.scales.stride(0), _stream()), "omg_quant_mx8")
    return out


def gemm_mx8(a: Mx8Tensor, w: Mx8Tensor, *, out_dtype: torch.dtype = torch.float16, bias: Optional[torch.Tensor] = None,
             residual: Optional[torch.Tensor] = None, act: int = L.ACT_NONE, out: Optional[torch.Tensor] = None,
             out_scale: float = 1.0, groups: int = 1, w_group_adapter: Optional[torch.Tensor] = None,
             n_per_adapter: Optional[int] = None, out_mx8: bool = False):
    """``out[M, N_out] = epi(dequant(a) @ dequant(w)^T)`` on the block-scaled fp8 MFMA (omg_gemm_mx8).

    ``out_mx8`` (GEGLU only, N % 256 == 0): return the result as an :class:`Mx8Tensor` — the next MX-fp8 Linear's operand,
    quantised in the epilogue, bit-identical to ``quant_mx8`` of the 16-bit result, which is never stored.

    With ``w_group_adapter`` (int32 [groups]) ``w`` holds ``n_adapters * n_per_adapter`` rows — the per-sample weight slots of
    the merged-LoRA mode stacked along the rows — and group g uses rows ``[id_g * n_per_adapter, (id_g + 1) * n_per_adapter)``."""
    M, K = a.rows, a.K
    assert w.K == K
    N = n_per_adapter if w_group_adapter is not None else w.rows
    n_out = N // 2 if act == L.ACT_GEGLU else N
    oq = None
    if out_mx8:
        assert out is None and act == L.ACT_GEGLU and N % 256 == 0
        oq = Mx8Tensor(torch.empty((M, n_out), dtype=torch.uint8, device=a.q.device),
                       torch.empty((n_out // 128, (M + 3) // 4 * 4), dtype=torch.int32, device=a.q.device), tuple(a.shape[:-1]) + (n_out,))
        out = oq.q
    elif out is None:
        out = torch.empty((M, n_out), dtype=out_dtype, device=a.q.device)
    assert out.stride(1) == 1 and out.shape == (M, n_out)
    g = L.GemmMx8Args()
    g.dtype = _DT[out_dtype] if out_mx8 else _dt(out)
    g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.a_scale, g.sa_ld = a.q.data_ptr(), a.q.stride(0), a.scales.data_ptr(), a.scales.stride(0)
    g.W, g.ldw, g.w_scale, g.sw_ld = w.q.data_ptr(), w.q.stride(0), w.scales.data_ptr(), w.scales.stride(0)
    g.groups, g.rows_per_group = groups, M // groups
    if w_group_adapter is not None:
        g.group_adapter = w_group_adapter.data_ptr()
        g.w_adapter_stride, g.sw_adapter_stride = N * w.q.stride(0), N
    g.bias = _p(bias)
    if residual is not None:
        assert residual.stride(1) == 1
        g.residual, g.ldr = residual.data_ptr(), residual.stride(0)
    g.act, g.out_scale = act, out_scale
    g.C, g.ldc = out.data_ptr(), out.stride(0)
    if oq is not None:
        g.c_scale, g.sc_ld = oq.scales.data_ptr(), oq.scales.stride(0)
    if _PROF is not None:
        t0 = _PROF.begin()
        L.check(L.lib().omg_gemm_mx8(C.byref(g), _stream()), "omg_gemm_mx8")
        _PROF.end("gemm_mx8", 2.0 * M * N * K, t0, ("mx8", M, N, K, 0, groups if w_group_adapter is not None else 1, act))
        return oq if oq is not None else out
    L.check(L.lib().omg_gemm_mx8(C.byref(g), _stream()), "omg_gemm_mx8")
    return oq if oq is not None else out


def conv2d(x1: torch.Tensor, w: torch.Tensor, ksize: int, *, stride: int = 1, upsample: bool = False,
           x2: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
           group_bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           out_scale: float = 1.0, act: int = L.ACT_NONE) -> torch.Tensor:
    """NHWC implicit-GEMM conv; ``w`` is ``[Cout, ksize*ksize*(C1+C2)]`` (see pack_conv_weight)."""
    _dev(x1)
    B, Hin, Win, C1 = x1.shape
    assert x1.is_contiguous()
    C2 = 0
    if x2 is not None:
        assert x2.is_contiguous() and x2.shape[:3] == x1.shape[:3]
        C2 = x2.shape[3]
    Cout = w.shape[0]
    assert w.is_contiguous() and w.shape[1] == ksize * ksize * (C1 + C2)
    Hl, Wl = (2 * Hin, 2 * Win) if upsample else (Hin, Win)
    pad = 1 if ksize == 3 else 0
    Hout = (Hl + 2 * pad - ksize) // stride + 1
    Wout = (Wl + 2 * pad - ksize) // stride + 1
    y = torch.empty((B, Hout, Wout, Cout), dtype=x1.dtype, device=x1.device)
    a = L.Conv2dArgs()
    a.dtype = _dt(x1)
    a.B, a.Hin, a.Win, a.C1, a.C2 = B, Hin, Win, C1, C2
    a.Hout, a.Wout, a.Cout = Hout, Wout, Cout
    a.ksize, a.stride, a.upsample = ksize, stride, int(upsample)
    a.X1, a.X2, a.W, a.bias = x1.data_ptr(), _p(x2), w.data_ptr(), _p(bias)
    if group_bias is not None:
        a.group_bias, a.ldgb = group_bias.data_ptr(), group_bias.stride(0)
    if residual is not None:
        assert residual.is_contiguous() and residual.shape == y.shape
        a.residual = residual.data_ptr()
    a.out_scale = out_scale
    a.act = act
    a.Y = y.data_ptr()
    if _PROF is not None:
        t0 = _PROF.begin()
        L.check(L.lib().omg_conv2d(C.byref(a), _stream()), "omg_conv2d")
        _PROF.end("gemm", 2.0 * B * Hout * Wout * Cout * ksize * ksize * (C1 + C2), t0, ("conv", B * Hout * Wout, Cout, ksize * ksize * (C1 + C2), 0, 1, 0))
        return y
    L.check(L.lib().omg_conv2d(C.byref(a), _stream()), "omg_conv2d")
    return y


class Mx8Map:
    """An MX-fp8 NHWC feature map: ``q`` uint8 [B, H, W, Cq] (OCP e4m3), ``scales`` int32 [Cq/128, B*H*W] (byte j of a dword = the
    E8M0 scale of channels 128 k + 32 j .. + 31 of that pixel) — what omg_groupnorm_mx8 writes and omg_conv2d_mx8 reads.
    Cq = the channel count rounded up to a multiple of 128; pad channels are zeros."""

    __slots__ = ("q", "scales", "shape", "dtype")

    def __init__(self, q: torch.Tensor, scales: torch.Tensor, dtype: torch.dtype):
        self.q, self.scales, self.shape, self.dtype = q, scales, tuple(q.shape), dtype

    @property
    def device(self):
        return self.q.device


def groupnorm_mx8(x1: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, *,
                  silu: bool = False, x2: Optional[torch.Tensor] = None) -> Mx8Map:
    """NHWC GroupNorm (+SiLU) of [x1 | x2] straight into MX-fp8 (the input of an MX-fp8 convolution): equal to quantising
    :func:`groupnorm`'s 16-bit output, which is never stored."""
    _dev(x1)
    assert x1.is_contiguous() and x1.dim() == 4
    B, H, W, C1 = x1.shape
    C2 = 0
    if x2 is not None:
        assert x2.is_contiguous() and x2.shape[:3] == x1.shape[:3]
        C2 = x2.shape[-1]
    Cc = C1 + C2
    if Cc % 32 != 0:
        raise L.OmgHipError("MX-fp8 feature maps need C % 32 == 0")
    Cq = (Cc + 127) // 128 * 128
    out = Mx8Map(torch.empty((B, H, W, Cq), dtype=torch.uint8, device=x1.device),
                 torch.empty((Cq // 128, B * H * W), dtype=torch.int32, device=x1.device), x1.dtype)
    ws = _gn_workspace(x1.device, B, groups, H * W)
    assert gamma.dtype == x1.dtype and beta.dtype == x1.dtype
    L.check(L.lib().omg_groupnorm_mx8(_dt(x1), x1.data_ptr(), C1, _p(x2), C2, B, H * W, groups, eps, gamma.data_ptr(), beta.data_ptr(),
                                      int(silu), ws.data_ptr(), out.q.data_ptr(), out.scales.data_ptr(), _stream()), "omg_groupnorm_mx8")
    return out


def conv2d_mx8(x: Mx8Map, w: "Mx8Tensor", *, bias: Optional[torch.Tensor] = None, group_bias: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, out_scale: float = 1.0, act: int = L.ACT_NONE) -> torch.Tensor:
    """3x3 / stride 1 / pad 1 convolution on the block-scaled fp8 MFMA (omg_conv2d_mx8).  ``w`` = quant_mx8 of the packed
    [Cout, 9*Cin] weight (pack_conv_weight); output, bias, per-sample bias and residual in ``x.dtype``."""
    _dev(x.q)
    B, H, W, Cin = x.shape
    Cout = w.rows
    assert w.K == 9 * Cin
    y = torch.empty((B, H, W, Cout), dtype=x.dtype, device=x.device)
    a = L.Conv2dMx8Args()
    a.dtype = _DT[x.dtype]
    a.B, a.H, a.W, a.Cin, a.Cout = B, H, W, Cin, Cout
    a.X, a.x_scale, a.Wq, a.w_scale = x.q.data_ptr(), x.scales.data_ptr(), w.q.data_ptr(), w.scales.data_ptr()
    a.sw_ld, a.act = w.scales.stride(0), act
    a.bias = _p(bias)
    if group_bias is not None:
        a.group_bias, a.ldgb = group_bias.data_ptr(), group_bias.stride(0)
    if residual is not None:
        assert residual.is_contiguous() and residual.shape == y.shape and residual.dtype == y.dtype
        a.residual = residual.data_ptr()
    a.out_scale = out_scale
    a.Y = y.data_ptr()
    if _PROF is not None:
        t0 = _PROF.begin()
        L.check(L.lib().omg_conv2d_mx8(C.byref(a), _stream()), "omg_conv2d_mx8")
        _PROF.end("gemm_mx8", 2.0 * B * H * W * Cout * 9 * Cin, t0, ("conv_mx8", B * H * W, Cout, 9 * Cin, 0, 1, 0))
        return y
    L.check(L.lib().omg_conv2d_mx8(C.byref(a), _stream()), "omg_conv2d_mx8")
    return y


def conv2d_f32(x: torch.Tensor, w: torch.Tensor, ksize: int, *, upsample: bool = False, bias: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 NHWC convolution on the f32-input MFMA (omg_conv2d_f32): the up blocks of the upcast VAE decode.
    ``w``: fp32 [Cout, ksize*ksize*Cin] (pack_conv_weight of the fp32 OIHW tensor)."""
    _dev(x)
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()
    B, Hin, Win, Cin = x.shape
    Cout = w.shape[0]
    assert w.shape[1] == ksize * ksize * Cin
    Ho, Wo = (2 * Hin, 2 * Win) if upsample else (Hin, Win)
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    a = L.Conv2dF32Args()
    a.B, a.Hin, a.Win, a.Cin, a.Hout, a.Wout, a.Cout, a.ksize, a.upsample = B, Hin, Win, Cin, Ho, Wo, Cout, ksize, int(upsample)
    a.X, a.W, a.Y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    if bias is not None:
        assert bias.dtype == torch.float32
        a.bias = bias.data_ptr()
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.is_contiguous() and residual.shape == y.shape
        a.residual = residual.data_ptr()
    if _PROF is not None:
        t0 = _PROF.begin()
        L.check(L.lib().omg_conv2d_f32(C.byref(a), _stream()), "omg_conv2d_f32")
        _PROF.end("gemm_f32", 2.0 * B * Ho * Wo * Cout * ksize * ksize * Cin, t0, ("conv_f32", B * Ho * Wo, Cout, ksize * ksize * Cin, 0, 1, 0))
        return y
    L.check(L.lib().omg_conv2d_f32(C.byref(a), _stream()), "omg_conv2d_f32")
    return y


def cast_f32(x: torch.Tensor) -> torch.Tensor:
    """16-bit -> fp32 copy (same shape / layout)."""
    _dev(x)
    assert x.is_contiguous() and x.numel() % 8 == 0
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    L.check(L.lib().omg_cast_f32(_dt(x), x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "omg_cast_f32")
    return y


def transpose_v(v: torch.Tensor, heads: int, nkv_pad: Optional[int] = None, out: Optional[torch.Tensor] = None,
                mfma_order: bool = True) -> torch.Tensor:
    """``v``: (B, Nkv, >=heads*64) view with unit inner stride -> Vt (B, heads, 64, Nkv_pad).  ``mfma_order`` (default): the key
    order :func:`attention` consumes ([0-3, 8-11, 4-7, 12-15] inside every 16 keys); False: a plain transpose."""
    _dev(v)
    B, Nkv = v.shape[0], v.shape[1]
    if nkv_pad is None:
        nkv_pad = (Nkv + 63) // 64 * 64
    vt = out if out is not None else torch.empty((B, heads, 64, nkv_pad), dtype=v.dtype, device=v.device)
    assert vt.is_contiguous() and tuple(vt.shape) == (B, heads, 64, nkv_pad)
    L.check(L.lib().omg_transpose_v(_dt(v), v.data_ptr(), v.stride(1), v.stride(0), B, heads, Nkv, nkv_pad,
                                    vt.data_ptr(), int(mfma_order), _stream()), "omg_transpose_v")
    return vt


class RowMajorV:
    """The V operand of a self-attention as the QKV projection wrote it — a (B, Nkv, heads*64) VIEW with unit inner stride — instead of the
    V^T image :func:`transpose_v` makes: :func:`attention` hands it to the kernel that transposes on the LDS read (omg_attn_args.V, ABI 6).
    More than 128 keys only (:func:`value_operand` chooses)."""

    def __init__(self, v: torch.Tensor):
        assert v.dim() == 3 and v.stride(2) == 1 and v.stride(1) % 8 == 0 and v.stride(0) % 8 == 0 and v.shape[1] > 128
        self.v = v


def value_operand(v: torch.Tensor, heads: int):
    """What :func:`attention` wants for ``v`` (B, Nkv, heads*64): the view itself above 128 keys when its strides allow (self-attention
    at 32 x 32 / 64 x 64), else the V^T image in MFMA key order."""
    if v.shape[1] > 128 and v.stride(2) == 1 and v.stride(1) % 8 == 0 and v.stride(0) % 8 == 0 and v.data_ptr() % 16 == 0:
        return RowMajorV(v)
    return transpose_v(v, heads)


def _attn_args(q, k, vt, heads, nkv, scale, qk_src, out, accumulate, out_scale) -> L.AttnArgs:
    a = L.AttnArgs()
    a.dtype = _dt(q)
    a.B, a.heads, a.Nq, a.Nkv = q.shape[0], heads, q.shape[1], nkv
    a.Q, a.ldq, a.q_bstride = q.data_ptr(), q.stride(1), q.stride(0)
    a.K, a.ldk, a.k_bstride = k.data_ptr(), k.stride(1), k.stride(0)
    if isinstance(vt, RowMajorV):
        a.V, a.ldv, a.v_bstride = vt.v.data_ptr(), vt.v.stride(1), vt.v.stride(0)
    elif vt is not None:
        a.Vt, a.Nkv_pad = vt.data_ptr(), vt.shape[3]
    a.qk_src = _p(qk_src)
    a.scale = scale
    a.accumulate = int(accumulate)
    a.out_scale = out_scale
    if out is not None:
        a.O, a.ldo, a.o_bstride = out.data_ptr(), out.stride(1), out.stride(0)
    return a


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, scale: float, *,
              qk_src: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              accumulate: bool = False, out_scale: float = 1.0) -> torch.Tensor:
    """Fused attention.  q: (B,Nq,>=heads*64) view, k: (B,Nkv,>=heads*64) view, vt from :func:`value_operand` (or transpose_v).

    ``qk_src`` (int32 device tensor [B]) implements the controller's probability replacement:
    sample b uses Q,K of sample qk_src[b] and its own V.
    """
    _dev(q)
    assert q.stride(2) == 1 and k.stride(2) == 1
    B, Nq = q.shape[0], q.shape[1]
    if out is None:
        out = torch.empty((B, Nq, heads * 64), dtype=q.dtype, device=q.device)
    a = _attn_args(q, k, vt, heads, k.shape[1], scale, qk_src, out, accumulate, out_scale)
    if _PROF is not None:
        t0 = _PROF.begin()
        L.check(L.lib().omg_attn_fwd(C.byref(a), _stream()), "omg_attn_fwd")
        _PROF.end("attn", 4.0 * B * heads * Nq * k.shape[1] * 64, t0, ("attn", B, heads, Nq, k.shape[1]))
        return out
    L.check(L.lib().omg_attn_fwd(C.byref(a), _stream()), "omg_attn_fwd")
    return out


def attn_probs(q: torch.Tensor, k: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """Materialised softmax probabilities (B*heads, Nq, Nkv) — protocol mode only."""
    _dev(q)
    B, Nq, Nkv = q.shape[0], q.shape[1], k.shape[1]
    p = torch.empty((B * heads, Nq, Nkv), dtype=q.dtype, device=q.device)
    a = _attn_args(q, k, None, heads, Nkv, scale, None, None, False, 1.0)
    L.check(L.lib().omg_attn_probs(C.byref(a), p.data_ptr(), _stream()), "omg_attn_probs")
    return p


def attn_apply_probs(p: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    _dev(p)
    B, Nkv = v.shape[0], v.shape[1]
    Nq = p.shape[1]
    assert p.is_contiguous() and p.shape[0] == B * heads and p.shape[2] == Nkv
    out = torch.empty((B, Nq, heads * 64), dtype=p.dtype, device=p.device)
    L.check(L.lib().omg_attn_apply_probs(_dt(p), p.data_ptr(), v.data_ptr(), v.stride(1), v.stride(0), B, heads, Nq,
                                         Nkv, out.data_ptr(), out.stride(1), out.stride(0), _stream()),
            "omg_attn_apply_probs")
    return out


def _gn_workspace(device, B: int, groups: int, HW: int) -> torch.Tensor:
    """Partial-sum scratch of one GroupNorm call.  Allocated PER CALL: a module-level buffer that grew on demand was, under
    hipGraph capture, born in the capturing engine's private pool — a graph of ANOTHER engine captured later kept the pointer, and
    when the first engine dropped its graphs (pointer epoch) the block was unmapped under the survivor's replay (memory access
    fault in a stage 1 -> stage 2 sequence at full size, round 3).  Inside a capture torch gives every call its own slot of the
    graph's pool; outside, the caching allocator makes this a free-list lookup."""
    n = int(L.lib().omg_groupnorm_ws_floats(B, groups, HW))
    return torch.empty(max(n, 1), dtype=torch.float32, device=device)


def groupnorm(x1: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, *,
              silu: bool = False, x2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NHWC GroupNorm (+SiLU) of the channel-concat [x1 | x2]; x: (B, H, W, C) or (B, HW, C)."""
    _dev(x1)
    assert x1.is_contiguous()
    B = x1.shape[0]
    C1 = x1.shape[-1]
    HW = x1.numel() // (B * C1)
    C2 = 0
    if x2 is not None:
        assert x2.is_contiguous()
        C2 = x2.shape[-1]
    y = torch.empty(x1.shape[:-1] + (C1 + C2,), dtype=x1.dtype, device=x1.device)
    ws = _gn_workspace(x1.device, B, groups, HW)
    assert gamma.dtype == x1.dtype and beta.dtype == x1.dtype
    L.check(L.lib().omg_groupnorm(_dt(x1, allow_f32=True), x1.data_ptr(), C1, _p(x2), C2, B, HW, groups, eps, gamma.data_ptr(),
                                  beta.data_ptr(), int(silu), ws.data_ptr(), y.data_ptr(), _stream()), "omg_groupnorm")
    return y


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    _dev(x)
    Cc = x.shape[-1]
    x2 = x.reshape(-1, Cc)
    assert x2.stride(1) == 1
    y = torch.empty((x2.shape[0], Cc), dtype=x.dtype, device=x.device)
    L.check(L.lib().omg_layernorm(_dt(x), x2.data_ptr(), x2.stride(0), x2.shape[0], Cc, eps, gamma.data_ptr(),
                                  beta.data_ptr(), y.data_ptr(), y.stride(0), _stream()), "omg_layernorm")
    return y.view(x.shape)


def layernorm_mx8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> "Mx8Tensor":
    """LayerNorm straight into MX-fp8 (bytes + block scales): the input of an MX-fp8 Linear; rows = all leading dims."""
    _dev(x)
    Cc = x.shape[-1]
    x2 = x.reshape(-1, Cc)
    assert x2.stride(1) == 1
    M = x2.shape[0]
    out = Mx8Tensor(torch.empty((M, Cc), dtype=torch.uint8, device=x.device),
                    torch.empty((Cc // 128, (M + 3) // 4 * 4), dtype=torch.int32, device=x.device), x.shape)
    L.check(L.lib().omg_layernorm_mx8(_dt(x), x2.data_ptr(), x2.stride(0), M, Cc, eps, gamma.data_ptr(), beta.data_ptr(),
                                      out.q.data_ptr(), out.q.stride(0), out.scales.data_ptr(), out.scales.stride(0), _stream()),
            "omg_layernorm_mx8")
    return out


def dwconv2d(x: torch.Tensor, taps: torch.Tensor, B: int, H: int, W: int, ksize: int, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Depthwise ``ksize x ksize`` convolution (stride 1, same padding) of NHWC rows ``x`` [B*H*W, C] (any row stride: a column slice
    of a wider buffer is fine); ``taps`` [ksize*ksize, C] tap-major.  omg_dwconv2d — LiteMLA's multi-scale aggregation."""
    _dev(x)
    M, Cc = x.shape
    assert M == B * H * W and x.stride(1) == 1 and taps.shape == (ksize * ksize, Cc) and taps.is_contiguous()
    y = torch.empty((M, Cc), dtype=x.dtype, device=x.device)
    L.check(L.lib().omg_dwconv2d(_dt(x), x.data_ptr(), x.stride(0), B, H, W, Cc, ksize, taps.data_ptr(), _p(bias), y.data_ptr(), y.stride(0), _stream()),
            "omg_dwconv2d")
    return y


def relu_linear_att(qkv: torch.Tensor, B: int, HW: int, groups: int, dim: int, eps: float) -> torch.Tensor:
    """EfficientViT's ReLU linear attention (fp32 inside) on NHWC rows ``qkv`` [B*HW, groups*3*dim] (group g: q | k | v at columns
    3 dim g) -> [B*HW, groups*dim].  omg_relu_linear_att."""
    _dev(qkv)
    M, Cc = qkv.shape
    assert M == B * HW and Cc == groups * 3 * dim and qkv.stride(1) == 1
    out = torch.empty((M, groups * dim), dtype=qkv.dtype, device=qkv.device)
    ws = torch.empty((int(L.lib().omg_relu_linear_att_ws_floats(B, groups, dim, HW)),), dtype=torch.float32, device=qkv.device)
    L.check(L.lib().omg_relu_linear_att(_dt(qkv), qkv.data_ptr(), qkv.stride(0), B, HW, groups, dim, eps, ws.data_ptr(), out.data_ptr(), out.stride(0),
                                        _stream()), "omg_relu_linear_att")
    return out


def conv_in(x_nchw: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    """NCHW latents (fp32 or `dtype`) -> NHWC features in `dtype`; w: [Cout][64] from pack_conv_in_weight."""
    _dev(x_nchw)
    assert x_nchw.is_contiguous()
    B, Cin, H, W = x_nchw.shape
    Cout = w.shape[0]
    assert w.shape[1] == 64 and w.is_contiguous()
    y = torch.empty((B, H, W, Cout), dtype=dtype, device=x_nchw.device)
    ws = torch.empty((B * H * W, 64), dtype=dtype, device=x_nchw.device)
    is_f32 = x_nchw.dtype == torch.float32
    assert is_f32 or x_nchw.dtype == dtype
    L.check(L.lib().omg_conv_in(_DT[dtype], x_nchw.data_ptr(), int(is_f32), B, Cin, H, W, w.data_ptr(), _p(bias), Cout,
                                ws.data_ptr(), y.data_ptr(), _stream()), "omg_conv_in")
    return y


def conv_out(x_nhwc: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor],
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NHWC features -> NCHW fp32; w: [Cout][3][3][Cin]."""
    _dev(x_nhwc)
    assert x_nhwc.is_contiguous()
    B, H, W, Cin = x_nhwc.shape
    Cout = w.shape[0]
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x_nhwc.device)
    assert out.is_contiguous() and out.dtype == torch.float32
    assert w.dtype == x_nhwc.dtype and (bias is None or bias.dtype == x_nhwc.dtype)
    L.check(L.lib().omg_conv_out(_dt(x_nhwc, allow_f32=True), x_nhwc.data_ptr(), B, H, W, Cin, w.data_ptr(), _p(bias), Cout,
                                 out.data_ptr(), _stream()), "omg_conv_out")
    return out


def softmax_rows_(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """In-place ``softmax(x * scale, dim=-1)`` of a 2-D tensor with unit inner stride (VAE mid-block attention scores)."""
    _dev(x)
    assert x.dim() == 2 and x.stride(1) == 1
    L.check(L.lib().omg_softmax_rows(_dt(x), x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), float(scale), _stream()),
            "omg_softmax_rows")
    return x


def channel_mix(x_nchw: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """fp32 NCHW 1x1 convolution with at most 8 channels on either side (AutoencoderKL.post_quant_conv)."""
    _dev(x_nchw)
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous()
    B, Cin, H, W = x_nchw.shape
    Cout = w.shape[0]
    assert w.numel() == Cout * Cin and (bias is None or (bias.dtype == torch.float32 and bias.numel() == Cout))
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x_nchw.device)
    L.check(L.lib().omg_channel_mix(x_nchw.data_ptr(), w.data_ptr(), _p(bias), B, Cin, Cout, H * W, y.data_ptr(), _stream()),
            "omg_channel_mix")
    return y


def timestep_embedding(t: torch.Tensor, dim: int, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """t: fp32 device tensor [n] -> [n, dim] (cos | sin), written into `out` (may be a column slice)."""
    _dev(t)
    assert t.dtype == torch.float32 and t.is_contiguous()
    n = t.numel()
    if out is None:
        out = torch.empty((n, dim), dtype=dtype, device=t.device)
    assert out.stride(1) == 1
    L.check(L.lib().omg_timestep_embedding(_DT[dtype], t.data_ptr(), n, dim, out.data_ptr(), out.stride(0), _stream()),
            "omg_timestep_embedding")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    _dev(x)
    assert x.is_contiguous()
    y = torch.empty_like(x)
    L.check(L.lib().omg_silu(_dt(x), x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "omg_silu")
    return y


def add_(y: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """y += a (same shape, contiguous)."""
    _dev(y)
    assert y.is_contiguous() and a.is_contiguous() and y.shape == a.shape and y.dtype == a.dtype
    L.check(L.lib().omg_add_inplace(_dt(y), y.data_ptr(), a.data_ptr(), y.numel(), _stream()), "omg_add_inplace")
    return y


def copy2d(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst[r, :cols] = src[r, :cols] for 2-D views with unit inner stride."""
    _dev(src)
    assert src.dim() == 2 and dst.dim() == 2 and src.shape == dst.shape and src.stride(1) == 1 and dst.stride(1) == 1
    L.check(L.lib().omg_copy2d(_dt(src), src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), src.shape[0],
                               src.shape[1], _stream()), "omg_copy2d")


def fuse_cfg_step(noise_pred: torch.Tensor, latents: torch.Tensor, coef: torch.Tensor, step_idx: torch.Tensor, *,
                  guidance_scale: float, fuse: bool = False, region_preds: Sequence[Optional[torch.Tensor]] = (),
                  masks: Sequence[Optional[torch.Tensor]] = (), model_input_next: Optional[torch.Tensor] = None,
                  advance: bool = True, fused_noise_out: Optional[torch.Tensor] = None) -> None:
    """Region fusion + CFG + scheduler step + next model input (see omg_fuse_cfg_step)."""
    _dev(noise_pred)
    assert noise_pred.dtype == torch.float32 and noise_pred.is_contiguous() and noise_pred.shape[0] == 4
    assert latents.dtype == torch.float32 and latents.is_contiguous() and latents.shape[0] == 2
    assert coef.dtype == torch.float32 and coef.is_contiguous() and coef.shape[-1] == 4
    assert step_idx.dtype == torch.int32
    _, Cc, H, W = noise_pred.shape
    a = L.StepArgs()
    a.C, a.H, a.W = Cc, H, W
    a.n_concepts = len(masks)
    assert len(region_preds) == len(masks) <= L.MAX_CONCEPTS
    a.fuse = int(fuse)
    a.guidance_scale = guidance_scale
    a.noise_pred = noise_pred.data_ptr()
    hm = wm = 0
    for i, (r, m) in enumerate(zip(region_preds, masks)):
        if r is not None:
            assert r.dtype == torch.float32 and r.is_contiguous() and r.shape == (2, Cc, H, W)
            a.region_pred[i] = r.data_ptr()
        if m is not None:
            assert m.dtype == torch.float32 and m.is_contiguous() and m.dim() == 2
            if hm:
                assert (hm, wm) == tuple(m.shape), "all masks must share one resolution"
            hm, wm = m.shape
            a.masks[i] = m.data_ptr()
    a.Hm, a.Wm = (hm, wm) if hm else (H, W)
    a.coef, a.step_idx, a.advance = coef.data_ptr(), step_idx.data_ptr(), int(advance)
    a.latents = latents.data_ptr()
    if model_input_next is not None:
        assert model_input_next.is_contiguous() and model_input_next.shape == (4, Cc, H, W)
        a.model_input_next = model_input_next.data_ptr()
        a.out_dtype = L.OMG_F32 if model_input_next.dtype == torch.float32 else _dt(model_input_next)
    if fused_noise_out is not None:
        assert fused_noise_out.dtype == torch.float32 and fused_noise_out.is_contiguous()
        a.fused_noise_out = fused_noise_out.data_ptr()
    L.check(L.lib().omg_fuse_cfg_step(C.byref(a), _stream()), "omg_fuse_cfg_step")


def gather_step(table: torch.Tensor, step_idx: torch.Tensor, out: torch.Tensor) -> None:
    """out = table[step_idx] with the step index read ON THE DEVICE (table: (S, ...), out: (...))."""
    _dev(table)
    assert table.is_contiguous() and out.is_contiguous() and table.shape[1:] == out.shape and step_idx.dtype == torch.int32
    L.check(L.lib().omg_gather_step(_dt(table), table.data_ptr(), step_idx.data_ptr(), out.data_ptr(), out.numel(), _stream()),
            "omg_gather_step")


def scale_model_input(latents: torch.Tensor, coef_cin: torch.Tensor, out: torch.Tensor) -> None:
    """out[4,C,H,W] = cin * cat([latents]*2); coef_cin: 1-element fp32 device tensor."""
    _dev(latents)
    assert latents.dtype == torch.float32 and latents.is_contiguous() and out.is_contiguous()
    n = latents[0].numel()
    dt = L.OMG_F32 if out.dtype == torch.float32 else _dt(out)
    L.check(L.lib().omg_scale_model_input(dt, latents.data_ptr(), coef_cin.data_ptr(), n, out.data_ptr(), _stream()),
            "omg_scale_model_input")


# ------------------------------------------------------------------ host-side weight packing
def pack_conv_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """diffusers conv weight [Cout, Cin, kh, kw] -> [Cout, kh*kw*Cin] (K = (tap, cin)); pure layout."""
    co, ci, kh, kw = w_oihw.shape
    return w_oihw.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()


def pack_conv_in_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """conv_in weight [Cout, Cin, 3, 3] -> [Cout, 64]: (ky, kx, ci) order, zero padded (one GEMM K-slice)."""
    co, ci, kh, kw = w_oihw.shape
    out = torch.zeros((co, 64), dtype=w_oihw.dtype, device=w_oihw.device)
    out[:, : kh * kw * ci] = w_oihw.permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
    return out


def geglu_row_perm(n_total: int) -> torch.Tensor:
    """Row permutation that interleaves GEGLU value/gate rows in blocks of 32 (see gemm.hip epilogue).

    packed row p: block = p // 64, j = p % 64; j < 32 -> value row block*32 + j,
    else gate row n_total/2 + block*32 + (j - 32).
    """
    nh = n_total // 2
    assert nh % 32 == 0
    p = torch.arange(n_total)
    blk, j = p // 64, p % 64
    return torch.where(j < 32, blk * 32 + j, nh + blk * 32 + (j - 32))
