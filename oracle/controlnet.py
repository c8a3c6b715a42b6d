"""ORACLE (test infrastructure): fp32 torch-CPU restatement of the SDXL ``ControlNetModel.forward`` that the
reference calls at src/pipelines/lora_pipeline.py:519-536 (pose/depth/canny ControlNet on the main pass) and
src/pipelines/instantid_pipeline.py:580-589, :638-648 (``controlnet2`` / IdentityNet on the concept pass).

Third-party arithmetic (``diffusers==0.25.0`` ``models.controlnet.ControlNetModel`` +
``ControlNetConditioningEmbedding``; not vendored, not installable here) restated from its published algorithm:
an encoder copy of the UNet (conv_in, time/add embeddings, the three down blocks, the mid block) whose input is
``conv_in(sample) + cond_embedding(controlnet_cond)`` and whose 9 + 1 feature maps go through 1x1 "zero" convolutions
and are multiplied by ``conditioning_scale``.  PARITY UNPINNED (no reference vectors exist); anchors: the parameter
count of the full-width topology (1,251 M, SURVEY.md BASELINE table says 1.243 B from the analytic model without the
conditioning embedding) and the state-dict key layout.  Reuses the building blocks of oracle/unet.py.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import unet as ou

COND_CHANNELS = (16, 32, 96, 256)


def param_shapes(cfg: ou.UNetConfig, cond_in: int = 3) -> Dict[str, Tuple[int, ...]]:
    full = ou.param_shapes(cfg)
    keep = {k: v for k, v in full.items() if k.startswith(("conv_in.", "time_embedding.", "add_embedding.", "down_blocks.", "mid_block."))}
    c0 = cfg.block_out_channels[0]

    def conv(name, i, o, k):
        keep[name + ".weight"] = (o, i, k, k)
        keep[name + ".bias"] = (o,)

    conv("controlnet_cond_embedding.conv_in", cond_in, COND_CHANNELS[0], 3)
    n = 0
    for i in range(len(COND_CHANNELS) - 1):
        conv(f"controlnet_cond_embedding.blocks.{n}", COND_CHANNELS[i], COND_CHANNELS[i], 3); n += 1
        conv(f"controlnet_cond_embedding.blocks.{n}", COND_CHANNELS[i], COND_CHANNELS[i + 1], 3); n += 1
    conv("controlnet_cond_embedding.conv_out", COND_CHANNELS[-1], c0, 3)
    # zero convs: one per skip tensor of the encoder, in order
    chans = [c0]
    nb = len(cfg.block_out_channels)
    for i, c in enumerate(cfg.block_out_channels):
        chans += [c] * cfg.layers_per_block
        if i != nb - 1:
            chans.append(c)
    for i, c in enumerate(chans):
        conv(f"controlnet_down_blocks.{i}", c, c, 1)
    conv("controlnet_mid_block", cfg.block_out_channels[-1], cfg.block_out_channels[-1], 1)
    return keep


def init_state_dict(cfg: ou.UNetConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Synthetic weights; the 'zero' convolutions get small random values so that parity tests see them."""
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
        sd[k] = w.to(dtype).float()
    return sd


def cond_embedding(sd, cond: torch.Tensor) -> torch.Tensor:
    """ControlNetConditioningEmbedding: conv_in, SiLU, 6 x (conv, SiLU) with stride 2 on every second, conv_out."""
    p = "controlnet_cond_embedding"
    h = F.silu(F.conv2d(cond.float(), sd[p + ".conv_in.weight"], sd[p + ".conv_in.bias"], padding=1))
    for n in range(2 * (len(COND_CHANNELS) - 1)):
        h = F.silu(F.conv2d(h, sd[f"{p}.blocks.{n}.weight"], sd[f"{p}.blocks.{n}.bias"], padding=1, stride=2 if n % 2 else 1))
    return F.conv2d(h, sd[p + ".conv_out.weight"], sd[p + ".conv_out.bias"], padding=1)


def controlnet_forward(sd, cfg: ou.UNetConfig, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale,
                       text_embeds, time_ids, attn_fn: ou.AttnFn = ou.plain_attention, guess_mode: bool = False) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """``guess_mode`` (diffusers 0.25.0 ControlNetModel.forward, "6. scaling", with ``global_pool_conditions`` False — recalled, third-party): the
    residuals are scaled by ``logspace(-1, 0, n_down + 1) * conditioning_scale`` (0.1 for the shallowest skip ... 1.0 for the mid block) instead of
    by ``conditioning_scale`` alone.  Reached from lora_pipeline.py:519-528 when the caller passes ``guess_mode=True``."""
    B = sample.shape[0]
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1)
    if t.numel() == 1:
        t = t.expand(B)
    c0 = cfg.block_out_channels[0]
    emb = ou._linear(sd, "time_embedding.linear_2", F.silu(ou._linear(sd, "time_embedding.linear_1", ou.timestep_embedding(t, c0))))
    time_embeds = ou.timestep_embedding(time_ids.float().flatten(), cfg.addition_time_embed_dim).reshape(B, -1)
    add = torch.cat([text_embeds.float(), time_embeds], dim=-1)
    emb = emb + ou._linear(sd, "add_embedding.linear_2", F.silu(ou._linear(sd, "add_embedding.linear_1", add)))
    ctx = encoder_hidden_states.float()
    h = ou._conv(sd, "conv_in", sample.float()) + cond_embedding(sd, controlnet_cond)
    skips = [h]
    nb = len(cfg.block_out_channels)
    for i, typ in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            h = ou.resnet_block(sd, f"down_blocks.{i}.resnets.{j}", cfg, h, emb)
            if typ == "CrossAttnDownBlock2D":
                h = ou.transformer_2d(sd, f"down_blocks.{i}.attentions.{j}", cfg, cfg.attention_head_dim[i],
                                      cfg.transformer_layers_per_block[i], h, ctx, attn_fn)
            skips.append(h)
        if i != nb - 1:
            h = ou._conv(sd, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2)
            skips.append(h)
    h = ou.resnet_block(sd, "mid_block.resnets.0", cfg, h, emb)
    h = ou.transformer_2d(sd, "mid_block.attentions.0", cfg, cfg.attention_head_dim[-1], cfg.transformer_layers_per_block[-1], h, ctx, attn_fn)
    h = ou.resnet_block(sd, "mid_block.resnets.1", cfg, h, emb)
    scales = guess_mode_scales(len(skips), conditioning_scale) if guess_mode else [conditioning_scale] * (len(skips) + 1)
    down = [F.conv2d(s, sd[f"controlnet_down_blocks.{i}.weight"], sd[f"controlnet_down_blocks.{i}.bias"]) * scales[i]
            for i, s in enumerate(skips)]
    mid = F.conv2d(h, sd["controlnet_mid_block.weight"], sd["controlnet_mid_block.bias"]) * scales[-1]
    return down, mid


def guess_mode_scales(n_down: int, conditioning_scale: float) -> List[float]:
    """``torch.logspace(-1, 0, n_down + 1) * conditioning_scale`` (fp32, as diffusers computes it)"""
    return [float(v) for v in (torch.logspace(-1, 0, n_down + 1) * conditioning_scale)]
