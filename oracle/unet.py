"""ORACLE (test infrastructure, never imported by omg_amd): fp32 torch-CPU restatement of the
SDXL ``UNet2DConditionModel.forward`` that OMG's pipelines call at
``src/pipelines/lora_pipeline.py:546-566`` and ``:592-599``.

The arithmetic itself lives in the un-vendored third-party dependency ``diffusers==0.25.0``
(``requirements.txt:5``), absent from /root/reference and not installable here, so this file
restates the published algorithm of these upstream symbols:
``models.unet_2d_condition.UNet2DConditionModel.forward``, ``unet_2d_blocks.{DownBlock2D,
CrossAttnDownBlock2D, UNetMidBlock2DCrossAttn, CrossAttnUpBlock2D, UpBlock2D}``,
``resnet.{ResnetBlock2D, Downsample2D, Upsample2D}``, ``transformer_2d.Transformer2DModel``
(use_linear_projection=True), ``attention.{BasicTransformerBlock, FeedForward, GEGLU}``,
``embeddings.{Timesteps, TimestepEmbedding}``.

PARITY UNPINNED for this file: the reference ships no tests, golden vectors or weights for the
UNet and diffusers cannot be imported, so the only anchors are (i) the parameter count of the
full-width topology (2,567,463,684 — the published SDXL-base UNet size, checked in
tests/test_oracle.py) and (ii) the reference's own call sites.  State-dict keys follow the
diffusers layout so that a real checkpoint would load.

Storage-precision emulation (round 4): inside ``with oracle.precision.rounding(torch.float16):`` every op's output is rounded to
fp16 as the reference's fp16 eager execution does (fp32 accumulation inside the op): the "fp16 oracle" of DESIGN §3.

Everything is functional over a plain ``dict[str, Tensor]`` (NCHW, fp32); the attention inner
step is delegated to a pluggable ``attn_fn`` so that the reference's processor + controller
sequence (oracle/attention.py) can be injected exactly where diffusers would call it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .precision import r as _r          # identity unless `with oracle.precision.rounding(torch.float16)` (storage-precision emulation)

Tensor = torch.Tensor


@dataclass
class UNetConfig:
    """Subset of the diffusers UNet2DConditionModel config that SDXL-base uses."""
    in_channels: int = 4
    out_channels: int = 4
    sample_size: int = 128
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    attention_head_dim: Tuple[int, ...] = (5, 10, 20)   # = number of heads per block (diffusers naming quirk)
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    time_cond_proj_dim: Optional[int] = None

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @staticmethod
    def sdxl() -> "UNetConfig":
        return UNetConfig()

    @staticmethod
    def tiny() -> "UNetConfig":
        """Same topology, 1/5 width, fewer transformer layers: runs in well under a second on CPU."""
        return UNetConfig(sample_size=16, block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2),
                          attention_head_dim=(1, 2, 4), cross_attention_dim=128, addition_time_embed_dim=32,
                          projection_class_embeddings_input_dim=64 + 6 * 32)


# --------------------------------------------------------------------------- parameter shapes
def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """All parameters of the topology with diffusers state-dict key names."""
    s: Dict[str, Tuple[int, ...]] = {}
    c0 = cfg.block_out_channels[0]
    ted = cfg.time_embed_dim

    def lin(name, i, o, bias=True):
        s[name + ".weight"] = (o, i)
        if bias:
            s[name + ".bias"] = (o,)

    def conv(name, i, o, k):
        s[name + ".weight"] = (o, i, k, k)
        s[name + ".bias"] = (o,)

    def norm(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    def resnet(name, i, o):
        norm(name + ".norm1", i)
        conv(name + ".conv1", i, o, 3)
        lin(name + ".time_emb_proj", ted, o)
        norm(name + ".norm2", o)
        conv(name + ".conv2", o, o, 3)
        if i != o:
            conv(name + ".conv_shortcut", i, o, 1)

    def transformer(name, c, layers):
        norm(name + ".norm", c)
        lin(name + ".proj_in", c, c)
        for l in range(layers):
            b = f"{name}.transformer_blocks.{l}"
            norm(b + ".norm1", c)
            for a, kv in (("attn1", c), ("attn2", cfg.cross_attention_dim)):
                lin(f"{b}.{a}.to_q", c, c, bias=False)
                lin(f"{b}.{a}.to_k", kv, c, bias=False)
                lin(f"{b}.{a}.to_v", kv, c, bias=False)
                lin(f"{b}.{a}.to_out.0", c, c)
            norm(b + ".norm2", c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", c, 8 * c)
            lin(b + ".ff.net.2", 4 * c, c)
        lin(name + ".proj_out", c, c)

    conv("conv_in", cfg.in_channels, c0, 3)
    lin("time_embedding.linear_1", c0, ted)
    lin("time_embedding.linear_2", ted, ted)
    lin("add_embedding.linear_1", cfg.projection_class_embeddings_input_dim, ted)
    lin("add_embedding.linear_2", ted, ted)

    out_c = c0
    nblocks = len(cfg.block_out_channels)
    for i, typ in enumerate(cfg.down_block_types):
        in_c, out_c = out_c, cfg.block_out_channels[i]
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
            if typ == "CrossAttnDownBlock2D":
                transformer(f"down_blocks.{i}.attentions.{j}", out_c, cfg.transformer_layers_per_block[i])
        if i != nblocks - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    mid_c = cfg.block_out_channels[-1]
    resnet("mid_block.resnets.0", mid_c, mid_c)
    transformer("mid_block.attentions.0", mid_c, cfg.transformer_layers_per_block[-1])
    resnet("mid_block.resnets.1", mid_c, mid_c)

    rev = list(reversed(cfg.block_out_channels))
    rev_layers = list(reversed(cfg.transformer_layers_per_block))
    out_c = rev[0]
    for i, typ in enumerate(cfg.up_block_types):
        prev_c, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, nblocks - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = in_c if j == cfg.layers_per_block else out_c
            res_in = prev_c if j == 0 else out_c
            resnet(f"up_blocks.{i}.resnets.{j}", res_in + skip, out_c)
            if typ == "CrossAttnUpBlock2D":
                transformer(f"up_blocks.{i}.attentions.{j}", out_c, rev_layers[i])
        if i != nblocks - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    norm("conv_norm_out", c0)
    conv("conv_out", c0, cfg.out_channels, 3)
    return s


def init_state_dict(cfg: UNetConfig, seed: int = 0, dtype: torch.dtype = torch.float32,
                    qk_gain: float = 1.0) -> Dict[str, Tensor]:
    """Seeded synthetic weights (no SDXL checkpoint exists offline; SURVEY §8d recipe).

    Linear/conv ~ N(0, 1/fan_in); norms gamma ~ 1 + 0.1 N, beta ~ 0.1 N; to_q/to_k are scaled by
    ``qk_gain`` (logit std ~ qk_gain^2).  The default 1.0 keeps the 40-layer random network
    well-conditioned: at 2.0 the softmax rows are near one-hot and merely rounding q/k/v to fp16
    INSIDE THIS ORACLE moves the output by 0.08 (measured), i.e. the comparison would test chaos,
    not kernels.  Peaked / dominant-logit rows are exercised at kernel level instead
    (tests/test_kernels_gpu.py::test_attention).
    Values are rounded through ``dtype`` so the product (fp16/bf16) and the oracle (fp32) see the
    same numbers.
    """
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith(".weight") and len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            w = torch.randn(shp, generator=g) * fan_in ** -0.5
            if ".to_q." in k or ".to_k." in k:
                w = w * qk_gain
        elif k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            w = 0.1 * torch.randn(shp, generator=g)
        sd[k] = w.to(dtype).to(torch.float32)
    return sd


# --------------------------------------------------------------------------- building blocks
def timestep_embedding(t: Tensor, dim: int) -> Tensor:
    """diffusers embeddings.get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return _r(torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1))      # diffusers: computed in fp32, `.to(dtype=sample.dtype)`


def _linear(sd, name, x):
    return _r(F.linear(x, sd[name + ".weight"], sd.get(name + ".bias")))


def _conv(sd, name, x, stride=1, padding=1):
    return _r(F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding))


def _gn(sd, name, x, groups, eps):
    return _r(F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps))


def _ln(sd, name, x):
    return _r(F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5))


def _silu(x):
    return _r(F.silu(x))


def resnet_block(sd, name, cfg, x, temb):
    h = _conv(sd, name + ".conv1", _silu(_gn(sd, name + ".norm1", x, cfg.norm_num_groups, cfg.norm_eps)))
    h = _r(h + _linear(sd, name + ".time_emb_proj", _silu(temb))[:, :, None, None])
    h = _conv(sd, name + ".conv2", _silu(_gn(sd, name + ".norm2", h, cfg.norm_num_groups, cfg.norm_eps)))
    if (name + ".conv_shortcut.weight") in sd:
        x = _conv(sd, name + ".conv_shortcut", x, padding=0)
    return _r(x + h)


# attn_fn(name, heads, q, k, v, is_cross) -> (B, N, C); q/k/v are (B, N, C) projections
AttnFn = Callable[[str, int, Tensor, Tensor, Tensor, bool], Tensor]


def plain_attention(name: str, heads: int, q: Tensor, k: Tensor, v: Tensor, is_cross: bool) -> Tensor:
    """softmax(QK^T/sqrt(d)) V per head (what AttnProcessor2_0 / xformers compute)."""
    B, N, C = q.shape
    d = C // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)

    p = _r(torch.softmax(_r(split(q) @ split(k).transpose(-1, -2) * d ** -0.5), dim=-1))       # emulation: scores and probabilities are stored
    return _r(p @ split(v)).permute(0, 2, 1, 3).reshape(B, N, C)


def attention(sd, name, heads, x, ctx, attn_fn: AttnFn, lora=None):
    is_cross = ctx is not None
    if is_cross and getattr(attn_fn, "cross_override", None) is not None:
        if lora is not None:                                          # e.g. IP-Adapter: own K/V projections for the ip tokens
            return attn_fn.cross_override(sd, name, heads, x, ctx, lora=lora)
        return attn_fn.cross_override(sd, name, heads, x, ctx)
    src = ctx if is_cross else x

    def proj(n, inp):
        y = _linear(sd, f"{name}.{n}", inp)
        if lora is not None:
            y = _r(y + lora(f"{name}.{n}", inp))
        return y

    q, k, v = proj("to_q", x), proj("to_k", src), proj("to_v", src)
    o = attn_fn(name, heads, q, k, v, is_cross)
    return proj("to_out.0", o)


def transformer_block(sd, name, heads, x, ctx, attn_fn, lora=None):
    x = _r(x + attention(sd, name + ".attn1", heads, _ln(sd, name + ".norm1", x), None, attn_fn, lora))
    x = _r(x + attention(sd, name + ".attn2", heads, _ln(sd, name + ".norm2", x), ctx, attn_fn, lora))
    h = _ln(sd, name + ".norm3", x)
    ff = _linear(sd, name + ".ff.net.0.proj", h)
    if lora is not None:
        ff = _r(ff + lora(name + ".ff.net.0.proj", h))
    val, gate = ff.chunk(2, dim=-1)
    g = _r(val * _r(F.gelu(gate)))
    out = _linear(sd, name + ".ff.net.2", g)
    if lora is not None:
        out = _r(out + lora(name + ".ff.net.2", g))
    return _r(x + out)


def transformer_2d(sd, name, cfg, heads, layers, x, ctx, attn_fn, lora=None):
    B, C, H, W = x.shape
    res = x
    h = _gn(sd, name + ".norm", x, cfg.norm_num_groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    y = _linear(sd, name + ".proj_in", h)
    if lora is not None:
        y = _r(y + lora(name + ".proj_in", h))
    h = y
    for l in range(layers):
        h = transformer_block(sd, f"{name}.transformer_blocks.{l}", heads, h, ctx, attn_fn, lora)
    y = _linear(sd, name + ".proj_out", h)
    if lora is not None:
        y = _r(y + lora(name + ".proj_out", h))
    return _r(y.reshape(B, H, W, C).permute(0, 3, 1, 2) + res)


def unet_forward(sd: Dict[str, Tensor], cfg: UNetConfig, sample: Tensor, timestep, encoder_hidden_states: Tensor,
                 text_embeds: Tensor, time_ids: Tensor, attn_fn: AttnFn = plain_attention,
                 down_block_additional_residuals: Optional[Sequence[Tensor]] = None,
                 mid_block_additional_residual: Optional[Tensor] = None, lora=None,
                 taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """fp32 NCHW forward.  ``lora(key, x) -> delta`` adds PEFT-style ``s*B(A(x))`` on Linear layers.
    ``taps`` (optional dict) receives intermediate activations by name for layer-wise parity."""
    B = sample.shape[0]
    sample = sample.float()
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1)
    if t.numel() == 1:
        t = t.expand(B)
    c0 = cfg.block_out_channels[0]
    emb = _linear(sd, "time_embedding.linear_2", _silu(_linear(sd, "time_embedding.linear_1", timestep_embedding(t, c0))))
    time_embeds = timestep_embedding(time_ids.float().flatten(), cfg.addition_time_embed_dim).reshape(B, -1)
    add = torch.cat([text_embeds.float(), time_embeds], dim=-1)
    emb = _r(emb + _linear(sd, "add_embedding.linear_2", _silu(_linear(sd, "add_embedding.linear_1", add))))
    ctx = encoder_hidden_states.float()
    sample = _r(sample)                                   # the pipeline hands the UNet latents in its own dtype (lora_pipeline.py:491-492)

    def tap(n, v):
        if taps is not None:
            taps[n] = v

    tap("emb", emb)

    h = _conv(sd, "conv_in", sample)
    tap("conv_in", h)
    skips: List[Tensor] = [h]
    nblocks = len(cfg.block_out_channels)
    for i, typ in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            h = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", cfg, h, emb)
            tap(f"down_blocks.{i}.resnets.{j}", h)
            if typ == "CrossAttnDownBlock2D":
                h = transformer_2d(sd, f"down_blocks.{i}.attentions.{j}", cfg, cfg.attention_head_dim[i],
                                   cfg.transformer_layers_per_block[i], h, ctx, attn_fn, lora)
                tap(f"down_blocks.{i}.attentions.{j}", h)
            skips.append(h)
        if i != nblocks - 1:
            h = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2)
            skips.append(h)
        tap(f"down_blocks.{i}", h)
    if down_block_additional_residuals is not None:
        skips = [_r(s + res_.float()) for s, res_ in zip(skips, down_block_additional_residuals)]
    h = resnet_block(sd, "mid_block.resnets.0", cfg, h, emb)
    tap("mid_block.resnets.0", h)
    h = transformer_2d(sd, "mid_block.attentions.0", cfg, cfg.attention_head_dim[-1],
                       cfg.transformer_layers_per_block[-1], h, ctx, attn_fn, lora)
    tap("mid_block.attentions.0", h)
    h = resnet_block(sd, "mid_block.resnets.1", cfg, h, emb)
    if mid_block_additional_residual is not None:
        h = _r(h + mid_block_additional_residual.float())
    tap("mid_block", h)
    rev_heads = list(reversed(cfg.attention_head_dim))
    rev_layers = list(reversed(cfg.transformer_layers_per_block))
    for i, typ in enumerate(cfg.up_block_types):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", cfg, h, emb)
            tap(f"up_blocks.{i}.resnets.{j}", h)
            if typ == "CrossAttnUpBlock2D":
                h = transformer_2d(sd, f"up_blocks.{i}.attentions.{j}", cfg, rev_heads[i], rev_layers[i], h, ctx,
                                   attn_fn, lora)
                tap(f"up_blocks.{i}.attentions.{j}", h)
        if i != nblocks - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", h)
        tap(f"up_blocks.{i}", h)
    h = _silu(_gn(sd, "conv_norm_out", h, cfg.norm_num_groups, cfg.norm_eps))
    return _conv(sd, "conv_out", h)


def count_attention_layers(cfg: UNetConfig) -> int:
    """Number of attn2 (= attn1) modules; the installer sets controller.num_att_layers = 2x this
    (src/pipelines/lora_pipeline.py:136-152). 70 for SDXL-base."""
    n = 0
    for i, typ in enumerate(cfg.down_block_types):
        if typ == "CrossAttnDownBlock2D":
            n += cfg.layers_per_block * cfg.transformer_layers_per_block[i]
    n += cfg.transformer_layers_per_block[-1]
    rev = list(reversed(cfg.transformer_layers_per_block))
    for i, typ in enumerate(cfg.up_block_types):
        if typ == "CrossAttnUpBlock2D":
            n += (cfg.layers_per_block + 1) * rev[i]
    return n


def make_lora(cfg: UNetConfig, names: Sequence[str], rank: int, seed: int, scale: float, dtype=torch.float32):
    """Synthetic LoRA adapters on Linear layers (A ~ N(0,1/r)... here A ~ N(0,1/in), B ~ N(0,1e-2)).

    Returns ``(weights, fn)``: weights[key] = (A [r,in], B [out,r]); fn(key, x) = scale * (x A^T) B^T.
    """
    shapes = param_shapes(cfg)
    g = torch.Generator().manual_seed(seed)
    w = {}
    for k in names:
        o, i = shapes[k + ".weight"]
        A = (torch.randn(rank, i, generator=g) * i ** -0.5).to(dtype).float()
        Bm = (torch.randn(o, rank, generator=g) * 0.1).to(dtype).float()
        w[k] = (A, Bm)

    def fn(key, x):
        if key not in w:
            return 0.0
        A, Bm = w[key]
        return _r(scale * _r(F.linear(_r(F.linear(x, A)), Bm)))          # PEFT: lora_B(lora_A(x)) * scaling, each an op of its own

    return w, fn


def lora_target_names(cfg: UNetConfig) -> List[str]:
    """All attention + feed-forward Linear layers (the synthetic LoRA coverage of SURVEY §8d)."""
    out = []
    for k in param_shapes(cfg):
        if k.endswith(".weight") and any(s in k for s in (".to_q.", ".to_k.", ".to_v.", ".to_out.0.", ".ff.net.0.proj.", ".ff.net.2.")):
            out.append(k[: -len(".weight")])
    return out
