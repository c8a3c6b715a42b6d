"""ORACLE (test infrastructure): fp32 torch-CPU restatement of the SDXL ``AutoencoderKL.decode`` the reference runs right
after the denoising loop (src/pipelines/lora_pipeline.py:635-661: ``needs_upcasting`` -> ``vae.to(float32)``,
``image = self.vae.decode(latents / self.vae.config.scaling_factor, return_dict=False)[0]``, then
``image_processor.postprocess``) — row N1 of SURVEY.md §8f.

Third-party arithmetic (``diffusers==0.25.0`` ``models.autoencoder_kl.AutoencoderKL`` / ``models.vae.Decoder`` /
``unet_2d_blocks.{UNetMidBlock2D,UpDecoderBlock2D}`` / ``resnet.{ResnetBlock2D,Upsample2D}`` /
``attention_processor.Attention``; not vendored, not installable here) restated from its published algorithm:

    z -> post_quant_conv (1x1, 4->4) -> conv_in (3x3, 4->C_top)
      -> mid: resnet, single-head attention over all H*W tokens (GroupNorm, q/k/v/out Linear, residual), resnet
      -> up blocks over reversed(block_out_channels): (layers_per_block + 1) resnets [+ nearest-2x upsample + conv3x3]
      -> GroupNorm + SiLU -> conv_out (3x3, C_0 -> 3)

ResnetBlock2D here has no time embedding: ``x + conv2(silu(norm2(conv1(silu(norm1(x))))))`` with a 1x1 ``conv_shortcut`` on x
when the channel count changes.  GroupNorm: 32 groups, eps 1e-6.  PARITY UNPINNED: no reference vectors or weights exist;
anchors are the state-dict key layout (diffusers') and the decoder's parameter count at the SDXL configuration
(49,490,179 for the decoder + 20 for post_quant_conv).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class VaeConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.13025

    @staticmethod
    def sdxl() -> "VaeConfig":
        return VaeConfig()

    @staticmethod
    def tiny() -> "VaeConfig":
        return VaeConfig(block_out_channels=(64, 128), layers_per_block=1)


def param_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    out: Dict[str, Tuple[int, ...]] = {}

    def conv(name, i, o, k):
        out[name + ".weight"] = (o, i, k, k)
        out[name + ".bias"] = (o,)

    def norm(name, c):
        out[name + ".weight"] = (c,)
        out[name + ".bias"] = (c,)

    def lin(name, i, o):
        out[name + ".weight"] = (o, i)
        out[name + ".bias"] = (o,)

    def resnet(name, i, o):
        norm(name + ".norm1", i); conv(name + ".conv1", i, o, 3)
        norm(name + ".norm2", o); conv(name + ".conv2", o, o, 3)
        if i != o:
            conv(name + ".conv_shortcut", i, o, 1)

    conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    rev = list(reversed(cfg.block_out_channels))
    top = rev[0]
    conv("decoder.conv_in", cfg.latent_channels, top, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for p in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(f"{a}.{p}", top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    prev = top
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
        prev = c
    norm("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", rev[-1], cfg.out_channels, 3)
    return out


def init_state_dict(cfg: VaeConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Synthetic weights: conv / linear ~ N(0, 1/fan_in), norm gamma ~ 1 + 0.1 N, biases ~ 0.1 N."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith(".weight") and len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            w = torch.randn(shp, generator=g) * fan_in ** -0.5
        elif k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            w = 0.1 * torch.randn(shp, generator=g)
        sd[k] = w.to(dtype)
    return sd


def _conv(sd, name, x, padding):
    return F.conv2d(x, sd[name + ".weight"].float(), sd[name + ".bias"].float(), padding=padding)


def _gn(sd, name, x, cfg):
    return F.group_norm(x, cfg.norm_num_groups, sd[name + ".weight"].float(), sd[name + ".bias"].float(), cfg.norm_eps)


def resnet_block(sd, name, cfg, x):
    h = _conv(sd, name + ".conv1", F.silu(_gn(sd, name + ".norm1", x, cfg)), 1)
    h = _conv(sd, name + ".conv2", F.silu(_gn(sd, name + ".norm2", h, cfg)), 1)
    if name + ".conv_shortcut.weight" in sd:
        x = _conv(sd, name + ".conv_shortcut", x, 0)
    return x + h


def mid_attention(sd, name, cfg, x):
    B, C, H, W = x.shape
    h = _gn(sd, name + ".group_norm", x, cfg).reshape(B, C, H * W).transpose(1, 2)        # (B, HW, C), one head of dim C
    q, k, v = (F.linear(h, sd[f"{name}.{p}.weight"].float(), sd[f"{name}.{p}.bias"].float()) for p in ("to_q", "to_k", "to_v"))
    p = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, dim=-1)
    o = F.linear(p @ v, sd[name + ".to_out.0.weight"].float(), sd[name + ".to_out.0.bias"].float())
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def decode(sd: Dict[str, Tensor], cfg: VaeConfig, z: Tensor, taps: Dict[str, Tensor] = None) -> Tensor:
    """``AutoencoderKL.decode(z)`` for ``z = latents / scaling_factor`` (the caller divides, as the reference does)."""
    x = _conv(sd, "post_quant_conv", z.float(), 0)
    x = _conv(sd, "decoder.conv_in", x, 1)
    x = resnet_block(sd, "decoder.mid_block.resnets.0", cfg, x)
    x = mid_attention(sd, "decoder.mid_block.attentions.0", cfg, x)
    x = resnet_block(sd, "decoder.mid_block.resnets.1", cfg, x)
    if taps is not None:
        taps["mid"] = x
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            x = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", cfg, x)
        if i != n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x, 1)
        if taps is not None:
            taps[f"up{i}"] = x
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, cfg))
    return _conv(sd, "decoder.conv_out", x, 1)


def postprocess(image: Tensor) -> Tensor:
    """``VaeImageProcessor.postprocess(..., output_type="pt")``: denormalise to [0, 1] (lora_pipeline.py:661)."""
    return (image / 2 + 0.5).clamp(0, 1)
