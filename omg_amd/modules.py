"""Parameter-holding building blocks of the MI355X UNet.

They are ``torch.nn.Module`` only so that ``state_dict()`` / ``load_state_dict()`` speak the
diffusers key layout ("state-dict keys must equal diffusers' so real
checkpoints load"); every ``forward`` dispatches to a HIP kernel through :mod:`omg_amd.ops`.
Activations are NHWC / token-major ``(B, H*W, C)`` everywhere inside the network.

Derived weight images (conv weights repacked to ``[Cout][ky][kx][Cin]``, GEGLU rows interleaved,
q|k|v concatenated) are built lazily from the canonical parameters and invalidated by
``load_state_dict`` / ``invalidate_packed``.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import _lib as L
from . import ops


class LoraState:
    """Per-forward LoRA selection: which adapter slot each sample of the batch uses (-1 = none)."""

    __slots__ = ("group_adapter", "groups", "merged")

    def __init__(self, group_adapter: torch.Tensor, groups: int, merged: bool = False):
        self.group_adapter = group_adapter   # int32 device tensor [groups]: LoRA slot (segment mode, -1 = none)
        self.groups = groups                 #   or merged-weight slot (merged mode, 0 = base weights)
        self.merged = merged


# Captured hipGraphs hold raw device pointers of the packed weight images and of the cached cross-attention K / V^T tensors.
# Every event that frees or re-allocates one of them (LoRA bank rebuild, ``load_state_dict``, ``.to()``, a cache eviction)
# advances this counter; the step engine compares it before each replay and drops its graphs when it moved
# (omg_amd/pipeline.py).
_POINTER_EPOCH = [0]


def pointer_epoch() -> int:
    return _POINTER_EPOCH[0]


def bump_weights_version(net) -> None:
    """``net.weights_version``: advances whenever the numbers a forward of ``net`` produces may change without its inputs changing — a state
    dict loaded, a layer class moved between 16-bit and MX-fp8.  Part of the stage cache's key (pipeline.StageCache; ADVICE r4: `id(unet)` alone
    let a precision change or a weight reload between the stage-1 and the stage-2 call resume from the other weights' latents)."""
    net.weights_version = getattr(net, "weights_version", 0) + 1


def bump_pointer_epoch() -> None:
    _POINTER_EPOCH[0] += 1


class _Packed(nn.Module):
    def __init__(self):
        super().__init__()
        self._packed = {}

    def invalidate_packed(self):
        if self._packed:
            bump_pointer_epoch()
        self._packed = {}

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)


class Linear(_Packed):
    """nn.Linear replacement; optional stacked LoRA adapters (``lora_down``/``lora_up`` buffers).

    ``forward(x, scale=None)`` accepts and ignores diffusers' legacy positional ``scale`` so that the
    reference processor's ``attn.to_q(hidden_states, *args)`` (lora_pipeline.py:98) works unchanged.
    """

    def __init__(self, in_features: int, out_features: int, bias: bool = True, dtype=torch.float16, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, dtype=dtype, device=device), requires_grad=False) if bias else None
        # LoRA bank (set by omg_amd.lora.LoraBank): [slots, r, in], [slots, out, r] (up already scaled)
        self.lora_down: Optional[torch.Tensor] = None
        self.lora_up: Optional[torch.Tensor] = None
        # merged mode: [1 + slots, out, in] = base weight followed by W + scale*B_s A_s per slot
        self.w_slots: Optional[torch.Tensor] = None
        self.lora_state: Optional[LoraState] = None   # set per forward by the UNet
        # MX-fp8 mode (UNet2DConditionModel.set_linear_precision): operands quantised to e4m3 with per-32 block scales, product on
        # the scaled MFMA (omg_gemm_mx8).  Quantised weight images live in ``_packed`` ("mx8_w", "mx8_slots").
        self.mx8 = False

    def invalidate_mx8_slots(self) -> None:
        if self._packed.pop("mx8_slots", None) is not None:
            bump_pointer_epoch()

    def _mx8_ok(self) -> bool:
        st = self.lora_state
        return self.in_features % 128 == 0 and not (st is not None and not st.merged and self.lora_down is not None)

    def _forward_mx8(self, x, residual, act):
        xq = x if isinstance(x, ops.Mx8Tensor) else ops.quant_mx8(x)
        shp = xq.shape
        r2 = residual.reshape(-1, self.out_features) if residual is not None else None
        st = self.lora_state
        dt = self.weight.dtype
        if st is not None and st.merged and self.w_slots is not None:
            if "mx8_slots" not in self._packed:
                self._packed["mx8_slots"] = ops.quant_mx8(self.w_slots.reshape(-1, self.in_features))
            y = ops.gemm_mx8(xq, self._packed["mx8_slots"], out_dtype=dt, bias=self.bias, residual=r2, act=act, groups=st.groups,
                             w_group_adapter=st.group_adapter, n_per_adapter=self.out_features)
        else:
            if "mx8_w" not in self._packed:
                self._packed["mx8_w"] = ops.quant_mx8(self.weight.data)
            y = ops.gemm_mx8(xq, self._packed["mx8_w"], out_dtype=dt, bias=self.bias, residual=r2, act=act)
        return y.view(*shp[:-1], y.shape[-1])

    def _lora(self, x2: torch.Tensor) -> Optional[ops.LoraSpec]:
        st = self.lora_state
        if st is None or st.merged or self.lora_down is None:
            return None
        t = ops.gemm(x2, self.lora_down, groups=st.groups, w_group_adapter=st.group_adapter)
        return ops.LoraSpec(t, self.lora_up, st.group_adapter)

    def forward(self, x: torch.Tensor, scale=None, *, residual: Optional[torch.Tensor] = None, act: int = L.ACT_NONE,
                group_bias: Optional[torch.Tensor] = None, groups: int = 1) -> torch.Tensor:
        if isinstance(x, ops.Mx8Tensor) or (self.mx8 and group_bias is None and self._mx8_ok()):
            if group_bias is not None or not self._mx8_ok():
                raise L.OmgHipError("an MX-fp8 activation reached a Linear layer that cannot run in MX-fp8")
            return self._forward_mx8(x, residual, act)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        r2 = residual.reshape(-1, self.out_features) if residual is not None else None
        st = self.lora_state
        if st is not None and st.merged and self.w_slots is not None:
            y = ops.gemm(x2, self.w_slots, bias=self.bias, residual=r2, act=act, groups=st.groups, w_group_adapter=st.group_adapter)
            return y.view(*shp[:-1], y.shape[-1])
        lora = self._lora(x2)
        if lora is not None:
            groups = self.lora_state.groups
            if x2.shape[0] % groups != 0:
                raise L.OmgHipError("LoRA groups do not divide the row count")
        y = ops.gemm(x2, self.weight, bias=self.bias, residual=r2, act=act, group_bias=group_bias, groups=groups, lora=lora)
        return y.view(*shp[:-1], y.shape[-1])


class GEGLU(_Packed):
    """diffusers attention.GEGLU: ``proj`` Linear(C -> 8C) then value * gelu(gate), fused in the GEMM epilogue."""

    def __init__(self, dim_in: int, dim_out: int, dtype=torch.float16, device=None):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2, dtype=dtype, device=device)

    def _packed_w(self):
        if "w" not in self._packed:
            perm = ops.geglu_row_perm(self.proj.out_features).to(self.proj.weight.device)
            self._packed["w"] = self.proj.weight.data[perm].contiguous()
            self._packed["b"] = self.proj.bias.data[perm].contiguous()
            if self.proj.lora_up is not None:
                self._packed["up"] = self.proj.lora_up[:, perm].contiguous()
            if self.proj.w_slots is not None:
                self._packed["w_slots"] = self.proj.w_slots[:, perm].contiguous()
        return self._packed

    def invalidate_packed(self):
        super().invalidate_packed()

    def _forward_mx8(self, x, out_mx8: bool = False):
        xq = x if isinstance(x, ops.Mx8Tensor) else ops.quant_mx8(x)
        shp = xq.shape
        pk = self._packed_w()
        st = self.proj.lora_state
        dt = self.proj.weight.dtype
        out_mx8 = out_mx8 and self.proj.out_features % 256 == 0
        if st is not None and st.merged and "w_slots" in pk:
            if "mx8_slots" not in pk:
                pk["mx8_slots"] = ops.quant_mx8(pk["w_slots"].reshape(-1, self.proj.in_features))
            y = ops.gemm_mx8(xq, pk["mx8_slots"], out_dtype=dt, bias=pk["b"], act=L.ACT_GEGLU, groups=st.groups,
                             w_group_adapter=st.group_adapter, n_per_adapter=self.proj.out_features, out_mx8=out_mx8)
        else:
            if "mx8_w" not in pk:
                pk["mx8_w"] = ops.quant_mx8(pk["w"])
            y = ops.gemm_mx8(xq, pk["mx8_w"], out_dtype=dt, bias=pk["b"], act=L.ACT_GEGLU, out_mx8=out_mx8)
        if isinstance(y, ops.Mx8Tensor):
            return y
        return y.view(*shp[:-1], y.shape[-1])

    def forward(self, x, out_mx8: bool = False):
        """``out_mx8``: the consumer is an MX-fp8 Linear — hand it its operand (bytes + block scales) straight from the epilogue."""
        if isinstance(x, ops.Mx8Tensor) or (self.proj.mx8 and self.proj._mx8_ok()):
            return self._forward_mx8(x, out_mx8)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        pk = self._packed_w()
        lora = None
        groups = 1
        st = self.proj.lora_state
        if st is not None and st.merged and "w_slots" in pk:
            y = ops.gemm(x2, pk["w_slots"], bias=pk["b"], act=L.ACT_GEGLU, groups=st.groups, w_group_adapter=st.group_adapter)
            return y.view(*shp[:-1], y.shape[-1])
        if st is not None and not st.merged and self.proj.lora_down is not None:
            t = ops.gemm(x2, self.proj.lora_down, groups=st.groups, w_group_adapter=st.group_adapter)
            lora = ops.LoraSpec(t, pk["up"], st.group_adapter)
            groups = st.groups
        y = ops.gemm(x2, pk["w"], bias=pk["b"], act=L.ACT_GEGLU, groups=groups, lora=lora)
        return y.view(*shp[:-1], y.shape[-1])


class Conv2d(_Packed):
    """3x3 (pad 1) or 1x1 conv on NHWC activations; canonical weight is diffusers' OIHW."""

    def __init__(self, cin: int, cout: int, ksize: int, stride: int = 1, dtype=torch.float16, device=None):
        super().__init__()
        self.cin, self.cout, self.ksize, self.stride = cin, cout, ksize, stride
        self.weight = nn.Parameter(torch.empty(cout, cin, ksize, ksize, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout, dtype=dtype, device=device), requires_grad=False)
        self.mx8 = False          # run on the block-scaled fp8 MFMA when fed an MX-fp8 feature map (UNet.set_conv_precision)

    def packed_weight(self) -> torch.Tensor:
        if "w" not in self._packed:
            self._packed["w"] = ops.pack_conv_weight(self.weight.data)
        return self._packed["w"]

    def mx8_ok(self) -> bool:
        """MX-fp8 eligibility: 3x3 / stride 1 on a 16-bit weight whose input channels are whole 32-element MX blocks."""
        return self.mx8 and self.ksize == 3 and self.stride == 1 and self.cin % 32 == 0 and self.weight.dtype != torch.float32

    def mx8_weight(self) -> "ops.Mx8Tensor":
        if "wq" not in self._packed:          # quantised once from the packed 16-bit weight; Cin padded with zeros to whole 128-wide K stages
            w = self.packed_weight()
            cq = (self.cin + 127) // 128 * 128
            if cq != self.cin:
                wp = torch.zeros((self.cout, 9, cq), dtype=w.dtype, device=w.device)
                wp[:, :, :self.cin] = w.view(self.cout, 9, self.cin)
                w = wp.view(self.cout, 9 * cq)
            self._packed["wq"] = ops.quant_mx8(w)
        return self._packed["wq"]

    def forward(self, x: torch.Tensor, *, x2: Optional[torch.Tensor] = None, upsample: bool = False,
                group_bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        if isinstance(x, ops.Mx8Map):                   # GroupNorm wrote the operand format of the fp8 MFMA directly
            if not self.mx8_ok() or x2 is not None or upsample:
                raise L.OmgHipError("an MX-fp8 feature map can only feed a 3x3 / stride 1 convolution in MX-fp8 mode")
            return ops.conv2d_mx8(x, self.mx8_weight(), bias=self.bias, group_bias=group_bias, residual=residual)
        if self.weight.dtype == torch.float32:          # fp32 storage (the up blocks of the upcast VAE decode): f32-input MFMA kernel
            if x2 is not None or group_bias is not None or self.stride != 1:
                raise L.OmgHipError("the fp32 convolution supports stride 1 without concat / per-sample bias (all the VAE decoder needs)")
            return ops.conv2d_f32(x, self.packed_weight(), self.ksize, upsample=upsample, bias=self.bias, residual=residual)
        return ops.conv2d(x, self.packed_weight(), self.ksize, stride=self.stride, upsample=upsample, x2=x2, bias=self.bias,
                          group_bias=group_bias, residual=residual)


class GroupNorm(nn.Module):
    def __init__(self, groups: int, channels: int, eps: float, dtype=torch.float16, device=None):
        super().__init__()
        self.groups, self.eps = groups, eps
        self.weight = nn.Parameter(torch.empty(channels, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(channels, dtype=dtype, device=device), requires_grad=False)

    def forward(self, x: torch.Tensor, *, x2: Optional[torch.Tensor] = None, silu: bool = False, mx8: bool = False):
        """``mx8=True``: the result as an :class:`omg_amd.ops.Mx8Map` for an MX-fp8 convolution (never stored in 16 bits)."""
        if mx8:
            return ops.groupnorm_mx8(x, self.weight, self.bias, self.groups, self.eps, silu=silu, x2=x2)
        return ops.groupnorm(x, self.weight, self.bias, self.groups, self.eps, silu=silu, x2=x2)


class LayerNorm(nn.Module):
    def __init__(self, channels: int, eps: float = 1e-5, dtype=torch.float16, device=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(channels, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(channels, dtype=dtype, device=device), requires_grad=False)

    def forward(self, x: torch.Tensor, mx8: bool = False):
        """``mx8=True``: the normalised rows as an :class:`omg_amd.ops.Mx8Tensor` for an MX-fp8 Linear (never stored in 16 bits)."""
        if mx8 and x.shape[-1] % 128 == 0:
            return ops.layernorm_mx8(x, self.weight, self.bias, self.eps)
        return ops.layernorm(x, self.weight, self.bias, self.eps)


class Dropout(nn.Module):
    """p = 0 at inference: identity (kept so that ``attn.to_out[1]`` exists, lora_pipeline.py:123)."""

    def forward(self, x):
        return x
