"""SDXL ``AutoencoderKL`` decoder on the MI355X kernels — the step right after the denoising loop (SURVEY.md §8f, row N1;
/root/reference src/pipelines/lora_pipeline.py:635-661: ``vae.decode(latents / scaling_factor)`` after the fp32 upcast,
then ``image_processor.postprocess``).

State-dict keys equal diffusers' (``post_quant_conv.*``, ``decoder.conv_in``, ``decoder.mid_block.{resnets,attentions}``,
``decoder.up_blocks.i.{resnets.j,upsamplers.0.conv}``, ``decoder.conv_norm_out``, ``decoder.conv_out``) so that
``omg_amd.loaders.load_model_weights(vae, "vae/diffusion_pytorch_model.safetensors", allow_extra=True)`` fills the decoder from
a full VAE file (the encoder / quant_conv entries are returned as ignored).

Every convolution is the implicit-GEMM MFMA kernel on NHWC activations (nearest-2x upsample folded into the consumer's
loader, residual adds in the GEMM epilogue), GroupNorm(+SiLU) is the UNet's kernel, and the mid-block attention — ONE head
of dimension C over all H*W tokens, which the head_dim-64 flash kernel cannot express — is two GEMMs around an in-place
row softmax (`omg_softmax_rows`): scores (HW x HW, 512 MB in 16-bit at a 128x128 latent) = q k^T / sqrt(C), out = P v.
Arithmetic: 16-bit storage / fp32 accumulation (the reference upcasts to fp32 because its fp16 VAE overflows; bf16 is the
safe choice with real SDXL weights, fp16 works with the synthetic ones).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from . import _lib as L
from . import ops
from .modules import Conv2d, GroupNorm, Linear


class VaeConfig:
    def __init__(self, latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, norm_eps=1e-6, scaling_factor=0.13025):
        self.latent_channels, self.out_channels = latent_channels, out_channels
        self.block_out_channels, self.layers_per_block = tuple(block_out_channels), layers_per_block
        self.norm_num_groups, self.norm_eps, self.scaling_factor = norm_num_groups, norm_eps, scaling_factor

    @staticmethod
    def sdxl() -> "VaeConfig":
        return VaeConfig()

    @staticmethod
    def tiny() -> "VaeConfig":
        return VaeConfig(block_out_channels=(64, 128), layers_per_block=1)


class _Resnet(nn.Module):
    """ResnetBlock2D without time embedding: x + conv2(silu(norm2(conv1(silu(norm1(x))))))  [+ 1x1 shortcut on x]."""

    def __init__(self, cin, cout, cfg, dtype, device):
        super().__init__()
        self.norm1 = GroupNorm(cfg.norm_num_groups, cin, cfg.norm_eps, dtype=dtype, device=device)
        self.conv1 = Conv2d(cin, cout, 3, dtype=dtype, device=device)
        self.norm2 = GroupNorm(cfg.norm_num_groups, cout, cfg.norm_eps, dtype=dtype, device=device)
        self.conv2 = Conv2d(cout, cout, 3, dtype=dtype, device=device)
        self.conv_shortcut = Conv2d(cin, cout, 1, dtype=dtype, device=device) if cin != cout else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = self.conv1(self.norm1(x, silu=True))
        sc = x if self.conv_shortcut is None else self.conv_shortcut(x)
        return self.conv2(self.norm2(h, silu=True), residual=sc)


class _MidAttention(nn.Module):
    def __init__(self, c, cfg, dtype, device):
        super().__init__()
        self.group_norm = GroupNorm(cfg.norm_num_groups, c, cfg.norm_eps, dtype=dtype, device=device)
        self.to_q = Linear(c, c, dtype=dtype, device=device)
        self.to_k = Linear(c, c, dtype=dtype, device=device)
        self.to_v = Linear(c, c, dtype=dtype, device=device)
        self.to_out = nn.ModuleList([Linear(c, c, dtype=dtype, device=device)])
        self._qkv = None

    def _apply(self, fn, *a, **k):
        self._qkv = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._qkv = None
        return super()._load_from_state_dict(*a, **k)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, H, W, C = x.shape
        HW = H * W
        if C % 64 or HW % 8:
            raise L.OmgHipError("VAE mid attention needs C % 64 == 0 and H*W % 8 == 0")
        if self._qkv is None:
            self._qkv = (torch.cat([self.to_q.weight.data, self.to_k.weight.data, self.to_v.weight.data]).contiguous(),
                         torch.cat([self.to_q.bias.data, self.to_k.bias.data, self.to_v.bias.data]).contiguous())
        h = self.group_norm(x).reshape(B * HW, C)
        qkv = ops.gemm(h, self._qkv[0], bias=self._qkv[1]).view(B, HW, 3 * C)
        vt = ops.transpose_v(qkv[:, :, 2 * C:], C // 64, mfma_order=False)     # (B, C/64, 64, HW_pad) == V^T [C][HW_pad]
        attn = torch.empty((B * HW, C), dtype=x.dtype, device=x.device)
        scores = torch.empty((HW, HW), dtype=x.dtype, device=x.device)
        for b in range(B):
            ops.gemm(qkv[b, :, :C], qkv[b, :, C:2 * C], out=scores)
            ops.softmax_rows_(scores, C ** -0.5)
            ops.gemm(scores, vt[b].reshape(C, -1)[:, :HW], out=attn[b * HW:(b + 1) * HW])
        y = self.to_out[0](attn, residual=x.reshape(B * HW, C))
        return y.view(B, H, W, C)


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, n, add_upsample, cfg, dtype, device):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if j == 0 else cout, cout, cfg, dtype, device) for j in range(n)])
        self.upsamplers = None
        if add_upsample:
            up = nn.Module()
            up.conv = Conv2d(cout, cout, 3, dtype=dtype, device=device)
            self.upsamplers = nn.ModuleList([up])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv(x, upsample=True)                       # nearest-2x folded into the conv's operand loader
        return x


class _Decoder(nn.Module):
    def __init__(self, cfg, dtype, device, up_dtype=None):
        super().__init__()
        up_dtype = up_dtype or dtype
        rev = list(reversed(cfg.block_out_channels))
        top = rev[0]
        self.conv_in_w = None
        self.conv_in = Conv2d(cfg.latent_channels, top, 3, dtype=dtype, device=device)      # parameters only; run by ops.conv_in
        self.mid_block = nn.Module()
        self.mid_block.resnets = nn.ModuleList([_Resnet(top, top, cfg, dtype, device), _Resnet(top, top, cfg, dtype, device)])
        self.mid_block.attentions = nn.ModuleList([_MidAttention(top, cfg, dtype, device)])
        blocks, prev = [], top
        for i, c in enumerate(rev):
            blocks.append(_UpBlock(prev, c, cfg.layers_per_block + 1, i != len(rev) - 1, cfg, up_dtype, device))
            prev = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = GroupNorm(cfg.norm_num_groups, rev[-1], cfg.norm_eps, dtype=up_dtype, device=device)
        self.conv_out = Conv2d(rev[-1], cfg.out_channels, 3, dtype=up_dtype, device=device)    # parameters only; run by ops.conv_out


class AutoencoderKLDecoder(nn.Module):
    """``decode(z)`` of diffusers' AutoencoderKL (decoder half + post_quant_conv)."""

    def __init__(self, cfg: Optional[VaeConfig] = None, dtype=torch.bfloat16, device="cuda", upcast: bool = False):
        """``upcast=True`` reproduces the reference's ``upcast_vae()`` decode (lora_pipeline.py:639-652; diffusers 0.25 with torch 2's
        attention processor [recalled]): post_quant_conv, conv_in and the mid block stay in ``dtype`` (the pipeline's fp16), the
        sample is cast to fp32 after the mid block and the up blocks, conv_norm_out and conv_out run in fp32 (fp32 weights, fp32
        activations, f32-input MFMA).  ``upcast=False``: everything in ``dtype`` (16-bit storage, fp32 accumulation) — 2-3x faster,
        the labelled throughput option; bf16 is then the overflow-safe choice with real SDXL weights."""
        super().__init__()
        self.config = cfg or VaeConfig.sdxl()
        self._dtype = dtype
        self.upcast = upcast
        c = self.config.latent_channels
        self.post_quant_conv = nn.Module()
        self.post_quant_conv.weight = nn.Parameter(torch.empty(c, c, 1, 1, dtype=torch.float32, device=device), requires_grad=False)
        self.post_quant_conv.bias = nn.Parameter(torch.empty(c, dtype=torch.float32, device=device), requires_grad=False)
        self.decoder = _Decoder(self.config, dtype, device, up_dtype=torch.float32 if upcast else None)
        self._packed = {}

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = {}
        return super().load_state_dict(*a, **k)

    def init_synthetic_(self, seed: int = 0) -> "AutoencoderKLDecoder":
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith(".weight") and p.dim() >= 2:
                w = torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32) * p[0].numel() ** -0.5
            elif name.endswith(".weight"):
                w = 1.0 + 0.1 * torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32)
            else:
                w = 0.1 * torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32)
            p.data.copy_(w.to(p.dtype))
        self._packed = {}
        return self

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """``z``: (B, 4, h, w) latents already divided by ``scaling_factor`` -> (B, 3, 8h, 8w) fp32 image in [-1, 1]-ish."""
        if not z.is_cuda:
            raise L.OmgHipError("AutoencoderKLDecoder.decode runs on the HIP kernels only (no CPU fallback)")
        d = self.decoder
        if "in" not in self._packed:
            self._packed["in"] = ops.pack_conv_in_weight(d.conv_in.weight.data)
            self._packed["out"] = d.conv_out.weight.data.permute(0, 2, 3, 1).contiguous()
            self._packed["pq"] = self.post_quant_conv.weight.data.reshape(self.config.latent_channels, -1).contiguous()
        x = ops.channel_mix(z.float().contiguous(), self._packed["pq"], self.post_quant_conv.bias.data)
        x = ops.conv_in(x, self._packed["in"], d.conv_in.bias.data, self._dtype)             # NCHW fp32 -> NHWC 16-bit
        x = d.mid_block.resnets[0](x)
        x = d.mid_block.attentions[0](x)
        x = d.mid_block.resnets[1](x)
        if self.upcast:
            x = ops.cast_f32(x)                                                 # `sample.to(upscale_dtype)` of diffusers' Decoder.forward
        for blk in d.up_blocks:
            x = blk(x)
        x = d.conv_norm_out(x, silu=True)
        return ops.conv_out(x, self._packed["out"], d.conv_out.bias.data)

    @torch.no_grad()
    def decode_latents(self, latents: torch.Tensor, postprocess: bool = True) -> torch.Tensor:
        """The reference's tail: ``vae.decode(latents / scaling_factor)`` then ``postprocess`` to [0, 1] (lora_pipeline.py:650-661)."""
        img = self.decode(latents.float() / self.config.scaling_factor)
        return (img / 2 + 0.5).clamp_(0, 1) if postprocess else img
