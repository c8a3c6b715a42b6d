"""EfficientViT's lightweight multi-scale linear attention (LiteMLA) on the HIP kernels — the segmentation hand-off between the
two stages (EfficientViT-SAM turns the stage-1 image into the region masks stage 2 needs).

Mirrors ``LiteMLA`` of /root/reference/src/efficientvit/models/nn/ops.py:335-455 — same constructor arguments, same state-dict keys
(``qkv.conv.weight``, ``aggreg.{i}.0.weight``, ``aggreg.{i}.1.weight``, ``proj.conv.weight``, ``proj.norm.*``), NCHW in / NCHW out —
so a checkpoint of the reference's module loads key for key.  MI355X-first layout: activations are NHWC rows inside;

  qkv 1x1 conv                     -> omg_gemm into the first 3T columns of ONE [B*HW, 3T(1+n)] buffer          (ops.py:366-373, :445)
  per scale: depthwise s x s conv  -> omg_dwconv2d reading that column slice                                      (ops.py:376-384)
             1x1 conv, 3*heads groups -> omg_gemm with the block-diagonal weight, writing the next 3T columns     (ops.py:385-391)
                                       (``torch.cat`` of :449 never happens; the zero blocks add exact zeros)
  relu_linear_att (fp32)           -> omg_relu_linear_att                                                         (ops.py:405-441)
  proj 1x1 conv + BatchNorm2d(eval) -> omg_gemm with the norm folded into weight and bias                         (ops.py:395-402, :452)

``forward(x, residual=True)`` adds the ``ResidualBlock`` shortcut of ``EfficientViTBlock.context_module`` (ops.py:470-481) in the
proj GEMM's epilogue.  Inference only (BatchNorm uses its running statistics); ``use_bias=False`` as in every EfficientViT model.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from . import ops


class _Conv(nn.Module):
    def __init__(self, shape, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(shape, dtype=dtype, device=device), requires_grad=False)


class _ConvLayer(nn.Module):
    """ConvLayer (ops.py:37-78) without dropout / activation: ``conv`` (+ ``norm`` = BatchNorm2d buffers when asked for)."""

    def __init__(self, cin, cout, norm: bool, dtype, device):
        super().__init__()
        self.conv = _Conv((cout, cin, 1, 1), dtype, device)
        self.norm = None
        if norm:
            self.norm = nn.Module()
            self.norm.weight = nn.Parameter(torch.ones(cout, dtype=dtype, device=device), requires_grad=False)
            self.norm.bias = nn.Parameter(torch.zeros(cout, dtype=dtype, device=device), requires_grad=False)
            self.norm.register_buffer("running_mean", torch.zeros(cout, dtype=torch.float32, device=device))
            self.norm.register_buffer("running_var", torch.ones(cout, dtype=torch.float32, device=device))
            self.norm.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long, device=device))
            self.norm.eps = 1e-5


class LiteMLA(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, heads: Optional[int] = None, heads_ratio: float = 1.0, dim: int = 8,
                 use_bias=False, norm=(None, "bn2d"), act_func=(None, None), kernel_func: str = "relu", scales: Sequence[int] = (5,),
                 eps: float = 1.0e-15, dtype=torch.float16, device=None):
        super().__init__()
        if use_bias not in (False, (False, False)) or tuple(act_func) != (None, None) or kernel_func != "relu" or norm[0] is not None \
                or norm[1] not in (None, "bn2d"):
            raise L.OmgHipError("LiteMLA: only the configuration EfficientViT uses is built (no bias, ReLU kernel, norm=(None, 'bn2d' | None))")
        if dim not in (8, 16, 32):
            raise L.OmgHipError("LiteMLA: dim must be 8, 16 or 32")
        self.eps, self.dim, self.scales = eps, dim, tuple(scales)
        self.heads = heads or int(in_channels // dim * heads_ratio)
        T = self.heads * dim
        self.total_dim, self.in_channels, self.out_channels = T, in_channels, out_channels
        self.qkv = _ConvLayer(in_channels, 3 * T, False, dtype, device)
        self.aggreg = nn.ModuleList([nn.ModuleList([_Conv((3 * T, 1, s, s), dtype, device), _Conv((3 * T, dim, 1, 1), dtype, device)])
                                     for s in self.scales])
        self.proj = _ConvLayer(T * (1 + len(self.scales)), out_channels, norm[1] == "bn2d", dtype, device)
        self._packed = {}

    # ------------------------------------------------------------------ derived weight images (built once)
    def _load_from_state_dict(self, *a, **k):
        self._packed = {}
        return super()._load_from_state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = {}
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def _pack(self):
        if self._packed:
            return self._packed
        T3, d = 3 * self.total_dim, self.dim
        dt = self.qkv.conv.weight.dtype
        pk = {"wqkv": self.qkv.conv.weight.data.reshape(T3, self.in_channels).contiguous(), "taps": [], "wbd": []}
        for (dw, pw), s in zip(self.aggreg, self.scales):
            pk["taps"].append(dw.weight.data.reshape(T3, s * s).t().contiguous())                    # [s*s, 3T] tap-major
            w = pw.weight.data.reshape(T3 // d, d, d)                                                 # [group, out, in]
            pk["wbd"].append(torch.block_diag(*w.float().unbind(0)).to(dt).contiguous())              # [3T, 3T], exact zeros off the blocks
        w = self.proj.conv.weight.data.reshape(self.out_channels, -1).float()
        bias = None
        if self.proj.norm is not None:                                                                # BatchNorm2d in eval mode = per-channel affine
            n = self.proj.norm
            a = n.weight.data.float() / torch.sqrt(n.running_var.float() + n.eps)
            w = w * a[:, None]
            bias = (n.bias.data.float() - n.running_mean.float() * a).to(dt)
        pk["wproj"], pk["bproj"] = w.to(dt).contiguous(), bias
        self._packed = pk
        return pk

    # ------------------------------------------------------------------ forward
    def forward_nhwc(self, x: torch.Tensor, residual: bool = False) -> torch.Tensor:
        """x (B, H, W, Cin) contiguous -> (B, H, W, Cout)."""
        B, H, W, Cin = x.shape
        assert Cin == self.in_channels and x.is_contiguous()
        pk = self._pack()
        M, T3, n = B * H * W, 3 * self.total_dim, len(self.scales)
        x2 = x.view(M, Cin)
        buf = torch.empty((M, T3 * (1 + n)), dtype=x.dtype, device=x.device)
        ops.gemm(x2, pk["wqkv"], out=buf[:, :T3])
        for i, s in enumerate(self.scales):
            t = ops.dwconv2d(buf[:, :T3], pk["taps"][i], B, H, W, s)
            ops.gemm(t, pk["wbd"][i], out=buf[:, T3 * (i + 1):T3 * (i + 2)])
        att = ops.relu_linear_att(buf, B, H * W, self.heads * (1 + n), self.dim, self.eps)
        if residual and self.out_channels != Cin:
            raise L.OmgHipError("LiteMLA: the residual shortcut needs out_channels == in_channels")
        y = ops.gemm(att, pk["wproj"], bias=pk["bproj"], residual=x2 if residual else None)
        return y.view(B, H, W, self.out_channels)

    def forward(self, x: torch.Tensor, residual: bool = False) -> torch.Tensor:
        """x (B, Cin, H, W) as the reference's module takes it -> (B, Cout, H, W) (a permuted view of the NHWC result)."""
        return self.forward_nhwc(x.permute(0, 2, 3, 1).contiguous(), residual=residual).permute(0, 3, 1, 2)
