"""IP-Adapter ``Resampler`` (InstantID's ``image_proj_model``) on the HIP kernels: one 512-d face embedding -> the 16
image-prompt tokens of the concept UNet's IP-Adapter branch (/root/reference src/ip_adapter/resampler.py:81-121, built at
src/pipelines/instantid_single_pieline.py:163-184, applied at :221-243).  Caller side of row A12; runs once per identity.

State-dict keys equal the reference module's, so ``load_state_dict(torch.load(ckpt)["image_proj"])`` works
(instantid_single_pieline.py:179-182).  Composition: LayerNorm and bias / residual GEMMs of the UNet path; the perceiver
attention (16 queries over 1 + 16 keys, 20 heads of 64) is the flash kernel; ``nn.GELU`` of the bias-free feed-forward runs
in the GEGLU epilogue with a constant-one value half, as in omg_amd/text_encoder.py.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as L
from . import ops
from .modules import LayerNorm, Linear


class _PerceiverAttention(nn.Module):
    def __init__(self, dim, heads, dtype, device):
        super().__init__()
        inner = 64 * heads
        self.heads = heads
        self.norm1 = LayerNorm(dim, 1e-5, dtype=dtype, device=device)
        self.norm2 = LayerNorm(dim, 1e-5, dtype=dtype, device=device)
        self.to_q = Linear(dim, inner, bias=False, dtype=dtype, device=device)
        self.to_kv = Linear(dim, 2 * inner, bias=False, dtype=dtype, device=device)
        self.to_out = Linear(inner, dim, bias=False, dtype=dtype, device=device)


class Resampler(nn.Module):
    def __init__(self, dim=1280, depth=4, dim_head=64, heads=20, num_queries=16, embedding_dim=512, output_dim=2048, ff_mult=4,
                 dtype=torch.float16, device="cuda"):
        super().__init__()
        if dim_head != 64:
            raise ValueError("dim_head must be 64 (the flash kernel's head size; InstantID uses 64)")
        self.dim, self.heads, self.num_queries, self._dtype = dim, heads, num_queries, dtype
        ff = int(dim * ff_mult)
        self.latents = nn.Parameter(torch.empty(1, num_queries, dim, dtype=dtype, device=device), requires_grad=False)
        self.proj_in = Linear(embedding_dim, dim, dtype=dtype, device=device)
        self.proj_out = Linear(dim, output_dim, dtype=dtype, device=device)
        self.norm_out = LayerNorm(output_dim, 1e-5, dtype=dtype, device=device)
        self.layers = nn.ModuleList([
            nn.ModuleList([_PerceiverAttention(dim, heads, dtype, device),
                           nn.ModuleList([LayerNorm(dim, 1e-5, dtype=dtype, device=device), Linear(dim, ff, bias=False, dtype=dtype, device=device),
                                          nn.Identity(), Linear(ff, dim, bias=False, dtype=dtype, device=device)])])
            for _ in range(depth)])
        self._ff_packed = None

    def load_state_dict(self, *a, **k):
        self._ff_packed = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._ff_packed = None
        return super()._apply(fn, *a, **k)

    def _pack(self):
        out = []
        for _, ff in self.layers:
            w1 = ff[1].weight.data
            f = w1.shape[0]
            perm = ops.geglu_row_perm(2 * f).to(w1.device)
            wg = torch.cat([torch.zeros_like(w1), w1])[perm].contiguous()
            bg = torch.cat([torch.ones(f, dtype=w1.dtype, device=w1.device), torch.zeros(f, dtype=w1.dtype, device=w1.device)])[perm].contiguous()
            out.append((wg, bg))
        self._ff_packed = out

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (B, n1, embedding_dim) -> (B, num_queries, output_dim)."""
        if not x.is_cuda:
            raise L.OmgHipError("Resampler runs on the HIP kernels only (no CPU fallback)")
        if self._ff_packed is None:
            self._pack()
        B, n1, _ = x.shape
        Q, D, heads = self.num_queries, self.dim, self.heads
        inner = 64 * heads
        xs = self.proj_in(x.to(self._dtype).reshape(B * n1, -1).contiguous())
        lat = self.latents.data.expand(B, Q, D).reshape(B * Q, D).contiguous()
        for (attn, ff), (wg, bg) in zip(self.layers, self._ff_packed):
            xn, ln = attn.norm1(xs), attn.norm2(lat)
            q = attn.to_q(ln).view(B, Q, inner)
            kv_in = torch.cat([xn.view(B, n1, D), ln.view(B, Q, D)], dim=1).reshape(B * (n1 + Q), D).contiguous()
            kv = attn.to_kv(kv_in).view(B, n1 + Q, 2 * inner)
            o = ops.attention(q, kv[:, :, :inner], ops.transpose_v(kv[:, :, inner:], heads), heads, 0.125)
            lat = attn.to_out(o.reshape(B * Q, inner), residual=lat)
            f = ops.gemm(ff[0](lat), wg, bias=bg, act=L.ACT_GEGLU)
            lat = ff[3](f, residual=lat)
        return self.norm_out(self.proj_out(lat)).view(B, Q, -1)
