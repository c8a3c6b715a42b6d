"""SDXL ``UNet2DConditionModel`` on hand-written gfx950 kernels — boundary B4 of SURVEY.md §8b.

Call contract kept from the reference's call sites (/root/reference
src/pipelines/lora_pipeline.py:546-566 and :592-599):

    unet(sample, timestep, encoder_hidden_states=, timestep_cond=None, cross_attention_kwargs=,
         down_block_additional_residuals=None, mid_block_additional_residual=None,
         added_cond_kwargs={"text_embeds", "time_ids"}, return_dict=False)[0]

plus the attributes the pipelines read: ``unet.config.{in_channels, sample_size,
time_cond_proj_dim, cross_attention_dim, block_out_channels}``, ``unet.dtype``, ``unet.device``,
``unet.attn_processors``, ``unet.set_attn_processor``.  The module tree and parameter names are
diffusers' (``down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight`` ...), so a real
SDXL checkpoint's ``state_dict`` loads with ``load_state_dict`` and the reference's installer
``revise_regionally_controlnet_forward`` can walk it.

MI355X-first layout: NCHW exists only at the boundary (conv_in reads NCHW latents, conv_out writes
NCHW fp32 noise); everything in between is NHWC == token-major (B, H*W, C), so convolutions are
implicit GEMMs over contiguous channel slices and the transformer blocks need no permutes.  The
skip-connection ``torch.cat`` and the nearest-2x upsample are folded into the consuming
convolution's / GroupNorm's operand loader.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _lib as L
from . import ops
from .attention import Attention, FusedAttnProcessor
from .modules import Conv2d, GEGLU, GroupNorm, LayerNorm, Linear, LoraState, bump_pointer_epoch, bump_weights_version


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    sample_size: int = 128
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    attention_head_dim: Tuple[int, ...] = (5, 10, 20)     # heads per block (diffusers naming)
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    time_cond_proj_dim: Optional[int] = None

    @staticmethod
    def sdxl() -> "UNetConfig":
        return UNetConfig()

    @staticmethod
    def tiny() -> "UNetConfig":
        return UNetConfig(sample_size=16, block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2),
                          attention_head_dim=(1, 2, 4), cross_attention_dim=128, addition_time_embed_dim=32,
                          projection_class_embeddings_input_dim=64 + 6 * 32)


class _Ctx:
    """Per-forward state threaded through the blocks."""

    __slots__ = ("silu_emb", "ehs", "B")

    def __init__(self, silu_emb, ehs, B):
        self.silu_emb, self.ehs, self.B = silu_emb, ehs, B


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim, dtype, device):
        super().__init__()
        self.linear_1 = Linear(cin, dim, dtype=dtype, device=device)
        self.linear_2 = Linear(dim, dim, dtype=dtype, device=device)

    def forward(self, x, residual=None):
        return self.linear_2(self.linear_1(x, act=L.ACT_SILU), residual=residual)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps, dtype, device):
        super().__init__()
        self.norm1 = GroupNorm(groups, cin, eps, dtype, device)
        self.conv1 = Conv2d(cin, cout, 3, dtype=dtype, device=device)
        self.time_emb_proj = Linear(temb_dim, cout, dtype=dtype, device=device)
        self.norm2 = GroupNorm(groups, cout, eps, dtype, device)
        self.conv2 = Conv2d(cout, cout, 3, dtype=dtype, device=device)
        self.conv_shortcut = Conv2d(cin, cout, 1, dtype=dtype, device=device) if cin != cout else None

    def forward(self, x, ctx: _Ctx, x2=None):
        """x (B,H,W,C1) [, x2 (B,H,W,C2): the skip tensor, consumed as channel-concat without copying]."""
        # MX-fp8 mode: the two GroupNorms write their convolution's operand format directly (bytes + per-pixel block scales)
        h = self.norm1(x, x2=x2, silu=True, mx8=self.conv1.mx8_ok())
        tproj = self.time_emb_proj(ctx.silu_emb)                     # (B, Cout)
        h = self.conv1(h, group_bias=tproj)
        h = self.norm2(h, silu=True, mx8=self.conv2.mx8_ok())
        if self.conv_shortcut is not None:
            sc = self.conv_shortcut(x, x2=x2)
        else:
            if x2 is not None:
                raise L.OmgHipError("concat input without a shortcut conv cannot happen in the SDXL topology")
            sc = x
        return self.conv2(h, residual=sc)


class FeedForward(nn.Module):
    def __init__(self, dim, dtype, device):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4, dtype, device), nn.Identity(), Linear(dim * 4, dim, dtype=dtype, device=device)])

    def forward(self, x, residual=None):
        # MX-fp8 mode: the GEGLU epilogue quantises for the output Linear (the 16-bit intermediate [rows, 4 dim] is never stored)
        return self.net[2](self.net[0](x, out_mx8=self.net[2].mx8 and self.net[2]._mx8_ok()), residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim, dtype, device):
        super().__init__()
        self.norm1 = LayerNorm(dim, dtype=dtype, device=device)
        self.attn1 = Attention(dim, None, heads, dtype=dtype, device=device)
        self.norm2 = LayerNorm(dim, dtype=dtype, device=device)
        self.attn2 = Attention(dim, cross_dim, heads, dtype=dtype, device=device)
        self.norm3 = LayerNorm(dim, dtype=dtype, device=device)
        self.ff = FeedForward(dim, dtype, device)

    @staticmethod
    def _attend(attn, x, h, ehs, kw):
        if getattr(attn.processor, "supports_fused_residual", False):
            return attn(x, encoder_hidden_states=ehs, residual=h, **kw)
        return attn(x, encoder_hidden_states=ehs, **kw) + h      # foreign processor: protocol only

    def forward(self, h, ehs, kw):
        # MX-fp8 mode: the three LayerNorms write their consumer GEMM's operand format directly (bytes + block scales)
        fused = getattr(self.attn1.processor, "supports_fused_residual", False) and getattr(self.attn2.processor, "supports_fused_residual", False)
        h = self._attend(self.attn1, self.norm1(h, mx8=fused and self.attn1.to_q.mx8 and not self.attn1._has_lora()), h, None, kw)
        h = self._attend(self.attn2, self.norm2(h, mx8=fused and self.attn2.to_q.mx8 and self.attn2.to_q._mx8_ok()), h, ehs, kw)
        return self.ff(self.norm3(h, mx8=self.ff.net[0].proj.mx8 and self.ff.net[0].proj._mx8_ok()), residual=h)


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, layers, cross_dim, groups, dtype, device):
        super().__init__()
        self.norm = GroupNorm(groups, dim, 1e-6, dtype, device)
        self.proj_in = Linear(dim, dim, dtype=dtype, device=device)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, cross_dim, dtype, device) for _ in range(layers)])
        self.proj_out = Linear(dim, dim, dtype=dtype, device=device)

    def forward(self, x, ehs, kw):
        B, H, W, C = x.shape
        h = self.proj_in(self.norm(x).view(B, H * W, C))
        for blk in self.transformer_blocks:
            h = blk(h, ehs, kw)
        return self.proj_out(h, residual=x.view(B, H * W, C)).view(B, H, W, C)


class Downsample2D(nn.Module):
    def __init__(self, c, dtype, device):
        super().__init__()
        self.conv = Conv2d(c, c, 3, stride=2, dtype=dtype, device=device)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c, dtype, device):
        super().__init__()
        self.conv = Conv2d(c, c, 3, dtype=dtype, device=device)

    def forward(self, x):
        return self.conv(x, upsample=True)      # nearest-2x folded into the conv's operand loader


class _DownBlock(nn.Module):
    def __init__(self, cfg, i, cin, cout, has_attn, add_down, dtype, device):
        super().__init__()
        ted = cfg.block_out_channels[0] * 4
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, ted, cfg.norm_num_groups, cfg.norm_eps, dtype, device)
                                      for j in range(cfg.layers_per_block)])
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, cfg.attention_head_dim[i], cfg.transformer_layers_per_block[i],
                                                                cfg.cross_attention_dim, cfg.norm_num_groups, dtype, device)
                                             for _ in range(cfg.layers_per_block)])
        self.has_attn = has_attn
        self.downsamplers = nn.ModuleList([Downsample2D(cout, dtype, device)]) if add_down else None

    def forward(self, h, ctx, kw, skips):
        for j, res in enumerate(self.resnets):
            h = res(h, ctx)
            if self.has_attn:
                h = self.attentions[j](h, ctx.ehs, kw)
            skips.append(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            skips.append(h)
        return h


class DownBlock2D(_DownBlock):
    pass


class CrossAttnDownBlock2D(_DownBlock):
    pass


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, cfg, dtype, device):
        super().__init__()
        c = cfg.block_out_channels[-1]
        ted = cfg.block_out_channels[0] * 4
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, ted, cfg.norm_num_groups, cfg.norm_eps, dtype, device) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, cfg.attention_head_dim[-1], cfg.transformer_layers_per_block[-1],
                                                            cfg.cross_attention_dim, cfg.norm_num_groups, dtype, device)])

    def forward(self, h, ctx, kw):
        h = self.resnets[0](h, ctx)
        h = self.attentions[0](h, ctx.ehs, kw)
        return self.resnets[1](h, ctx)


class _UpBlock(nn.Module):
    def __init__(self, cfg, heads, layers, prev_c, cout, in_c, has_attn, add_up, dtype, device):
        super().__init__()
        ted = cfg.block_out_channels[0] * 4
        n = cfg.layers_per_block + 1
        self.resnets = nn.ModuleList()
        for j in range(n):
            skip = in_c if j == n - 1 else cout
            rin = prev_c if j == 0 else cout
            self.resnets.append(ResnetBlock2D(rin + skip, cout, ted, cfg.norm_num_groups, cfg.norm_eps, dtype, device))
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, layers, cfg.cross_attention_dim, cfg.norm_num_groups, dtype, device)
                                             for _ in range(n)])
        self.has_attn = has_attn
        self.upsamplers = nn.ModuleList([Upsample2D(cout, dtype, device)]) if add_up else None

    def forward(self, h, ctx, kw, skips):
        for j, res in enumerate(self.resnets):
            h = res(h, ctx, x2=skips.pop())
            if self.has_attn:
                h = self.attentions[j](h, ctx.ehs, kw)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class UpBlock2D(_UpBlock):
    pass


class CrossAttnUpBlock2D(_UpBlock):
    pass


MX8_CLASSES = ("qkv", "attn_out", "cross_q", "cross_out", "geglu", "ff_out", "proj_in", "proj_out", "conv1", "conv2")
# round 4: the per-class sensitivity sweep (tools/mx8_sensitivity.py, profiles/r04_mx8_sensitivity.json: 50-step loop error
# of each class alone in fp8, of all but it, and of the mixes) found NO dominant class — every class alone costs rms 1.6e-2 ... 5.3e-2 of the
# latent rms and the ten add in quadrature to the 0.104 of the full mode (sqrt(sum of squares) = 0.11): the error is the e4m3 format's
# (3 mantissa bits on both operands of every product), not a few sensitive layers'.  Presets: "all" = the throughput mode of `--dtype fp8`;
# "safe" = the three quietest classes (50-step rms 3.1e-2, 21 % of the FLOPs on the fp8 MFMA) for callers that want the tighter bound.
MX8_PRESETS = {"all": MX8_CLASSES, "safe": ("cross_q", "cross_out", "ff_out"), "none": ()}


def mx8_class_of(name: str, module) -> Optional[str]:
    """The MX-fp8 layer class of a module of the UNet / ControlNet (None: never eligible — conv_in / conv_out, up / down-sampling and
    shortcut convolutions, the K / V projections of the text context, time / text embeddings: SURVEY §7.3 item 8)."""
    if isinstance(module, Linear) and ".attentions." in name:
        for suffix, cls in ((".attn1.to_q", "qkv"), (".attn1.to_k", "qkv"), (".attn1.to_v", "qkv"), (".attn1.to_out.0", "attn_out"),
                            (".attn2.to_q", "cross_q"), (".attn2.to_out.0", "cross_out"), (".ff.net.0.proj", "geglu"), (".ff.net.2", "ff_out"),
                            (".proj_in", "proj_in"), (".proj_out", "proj_out")):
            if name.endswith(suffix):
                return cls if module.in_features % 128 == 0 else None
        return None
    if isinstance(module, Conv2d) and ".resnets." in name and (name.endswith(".conv1") or name.endswith(".conv2")):
        return name[-5:]
    return None


def set_mx8_classes(net, classes, select=None) -> int:
    """Put exactly the layer classes in ``classes`` (names of MX8_CLASSES, or a preset name of MX8_PRESETS; empty = everything 16-bit) of ``net`` (UNet or ControlNet) on
    the block-scaled fp8 MFMA.  ``select(name, cls) -> bool`` (optional) narrows further, e.g. to keep the first / last transformer
    block of a resolution in 16 bits.  Producers follow their consumers (LayerNorm / GroupNorm + SiLU / the GEGLU epilogue write the
    MX-fp8 operand when the layer they feed is fp8).  Returns the number of fp8 layers."""
    if isinstance(classes, str):                      # a preset name ("all" | "safe" | "none") or one class name
        classes = MX8_PRESETS.get(classes, (classes,))
    classes = set(classes or ())
    unknown = classes - set(MX8_CLASSES)
    if unknown:
        raise ValueError(f"unknown MX-fp8 layer classes {sorted(unknown)}; known: {MX8_CLASSES}")
    changed, n_on = False, 0
    for name, m in net.named_modules():
        if not isinstance(m, (Linear, Conv2d)):
            continue
        cls = mx8_class_of(name, m)
        new = cls is not None and cls in classes and (select is None or bool(select(name, cls)))
        changed |= bool(getattr(m, "mx8", False)) != new
        m.mx8 = new
        n_on += int(new)
    if changed:      # step graphs captured under the other precision launch the other kernels: drop them (pipeline.run_step)
        bump_pointer_epoch()
        bump_weights_version(net)
    return n_on


class UNet2DConditionModel(nn.Module):
    def __init__(self, config: Optional[UNetConfig] = None, dtype: torch.dtype = torch.float16, device=None):
        super().__init__()
        cfg = config or UNetConfig.sdxl()
        self.config = cfg
        self._dtype = dtype
        c0 = cfg.block_out_channels[0]
        ted = c0 * 4
        if cfg.time_cond_proj_dim is not None:
            raise L.OmgHipError("time_cond_proj_dim (LCM guidance embedding) is not part of SDXL-base")
        self.conv_in = Conv2d(cfg.in_channels, c0, 3, dtype=dtype, device=device)
        self.time_embedding = TimestepEmbedding(c0, ted, dtype, device)
        self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, ted, dtype, device)
        nb = len(cfg.block_out_channels)
        # Registration order follows diffusers' UNet2DConditionModel.__init__, which creates BOTH block lists before the mid
        # block: named_modules() / ``attn_processors`` therefore enumerate down_blocks, up_blocks, mid_block — the order the
        # InstantID ``ip_adapter`` checkpoint is indexed by (instantid_single_pieline.py:186-213).
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
        cout = c0
        for i, typ in enumerate(cfg.down_block_types):
            cin, cout = cout, cfg.block_out_channels[i]
            cls = CrossAttnDownBlock2D if typ == "CrossAttnDownBlock2D" else DownBlock2D
            self.down_blocks.append(cls(cfg, i, cin, cout, typ == "CrossAttnDownBlock2D", i != nb - 1, dtype, device))
        self.mid_block = UNetMidBlock2DCrossAttn(cfg, dtype, device)
        rev = list(reversed(cfg.block_out_channels))
        rev_heads = list(reversed(cfg.attention_head_dim))
        rev_layers = list(reversed(cfg.transformer_layers_per_block))
        cout = rev[0]
        for i, typ in enumerate(cfg.up_block_types):
            prev_c, cout = cout, rev[i]
            in_c = rev[min(i + 1, nb - 1)]
            cls = CrossAttnUpBlock2D if typ == "CrossAttnUpBlock2D" else UpBlock2D
            self.up_blocks.append(cls(cfg, rev_heads[i], rev_layers[i], prev_c, cout, in_c, typ == "CrossAttnUpBlock2D", i != nb - 1, dtype, device))
        self.conv_norm_out = GroupNorm(cfg.norm_num_groups, c0, cfg.norm_eps, dtype, device)
        self.conv_out = Conv2d(c0, cfg.out_channels, 3, dtype=dtype, device=device)
        self._boundary = {}
        self._linears: Optional[List[Linear]] = None

    # ------------------------------------------------------------------ diffusers-style attributes
    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def attentions(self):
        for name, m in self.named_modules():
            if isinstance(m, Attention):
                yield name, m

    @property
    def attn_processors(self) -> Dict[str, object]:
        return {f"{name}.processor": m.processor for name, m in self.attentions()}

    def set_attn_processor(self, processor) -> None:
        if isinstance(processor, dict):
            for name, m in self.attentions():
                m.set_processor(processor[f"{name}.processor"])
        else:
            for _, m in self.attentions():
                m.set_processor(processor)

    def set_default_attn_processor(self) -> None:
        self.set_attn_processor(FusedAttnProcessor())

    def invalidate_packed(self) -> None:
        self._boundary = {}
        for m in self.modules():
            if m is not self and hasattr(m, "invalidate_packed"):
                m.invalidate_packed()

    def _apply(self, fn, *a, **k):
        self._boundary = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_packed()
        bump_weights_version(self)
        return r

    def init_synthetic_(self, seed: int = 0, qk_gain: float = 1.0) -> "UNet2DConditionModel":
        """Seeded random weights ON DEVICE (no SDXL checkpoint is available offline):
        Linear/conv ~ N(0, 1/fan_in), norm gamma ~ 1 + 0.1 N, biases ~ 0.1 N, to_q/to_k scaled by qk_gain."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith(".weight") and p.dim() >= 2:
                fan_in = p[0].numel()
                w = torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32) * fan_in ** -0.5
                if ".to_q." in name or ".to_k." in name:
                    w *= qk_gain
            elif name.endswith(".weight"):
                w = 1.0 + 0.1 * torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32)
            else:
                w = 0.1 * torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32)
            p.data.copy_(w.to(p.dtype))
        self.invalidate_packed()
        return self

    # ------------------------------------------------------------------ precision of the transformer Linear layers
    def set_linear_precision(self, mode: str = "fp16") -> None:
        """``"mx8"``: every Linear inside the transformer blocks whose K is a multiple of 128 (q|k|v, to_q, to_out, GEGLU, FF-out,
        proj_in / proj_out: 66 % of the UNet's FLOPs) runs on the block-scaled fp8 MFMA with OCP MX operands (omg_gemm_mx8);
        activations are quantised by the producing LayerNorm or by omg_quant_mx8.  Kept in 16 bits, as SURVEY §7.3 item 8 asks:
        conv_in / conv_out, every convolution, the time / text embeddings and their projections, the cross-attention K / V
        projections of the (constant) text context, all norms' statistics, attention itself.  ``"fp16"`` (the name covers bf16
        storage too) restores the 16-bit GEMMs."""
        if mode not in ("fp16", "mx8"):
            raise ValueError(mode)
        on = mode == "mx8"
        changed = False
        for name, m in self.named_modules():
            if isinstance(m, Linear):
                in_block = ".attentions." in name and not (name.endswith(".to_k") or name.endswith(".to_v")) or \
                           (".attentions." in name and ".attn1." in name)
                new = on and in_block and m.in_features % 128 == 0
                changed |= bool(getattr(m, "mx8", False)) != new
                m.mx8 = new
        self.linear_precision = mode
        if changed:      # step graphs captured under the other precision launch the other kernels: drop them (pipeline.run_step)
            bump_pointer_epoch()
            bump_weights_version(self)

    def set_conv_precision(self, mode: str = "fp16") -> None:
        """``"mx8"``: conv1 / conv2 of every ResnetBlock2D run as MX-fp8 implicit GEMMs (omg_conv2d_mx8) on the feature map their
        GroupNorm + SiLU writes in MX-fp8 (omg_groupnorm_mx8; 320- and 960-channel maps padded with zero channels to 384 / 1024).  Kept in 16 bits: conv_in / conv_out, the down / up-sampling
        convolutions and the 1x1 shortcuts (their inputs are residual-stream tensors no norm has bounded), GroupNorm statistics,
        the time-embedding projection added as per-sample bias, the skip connection added in the epilogue."""
        if mode not in ("fp16", "mx8"):
            raise ValueError(mode)
        changed = False
        for m in self.modules():
            if isinstance(m, ResnetBlock2D):
                changed |= bool(getattr(m.conv1, "mx8", False)) != (mode == "mx8")
                m.conv1.mx8 = m.conv2.mx8 = mode == "mx8"
        self.conv_precision = mode
        if changed:
            bump_pointer_epoch()
            bump_weights_version(self)

    def set_precision_classes(self, classes, select=None) -> int:
        """Per-class MX-fp8 map (:func:`set_mx8_classes`); ``set_linear_precision("mx8")`` + ``set_conv_precision("mx8")`` = every class."""
        if isinstance(classes, str):          # a preset name or one class name, as set_mx8_classes resolves it (set("none") = {'n','o','e'})
            classes = MX8_PRESETS.get(classes, (classes,))
        n = set_mx8_classes(self, classes, select)
        cl = set(classes or ())
        self.linear_precision = "mx8" if cl - {"conv1", "conv2"} else "fp16"
        self.conv_precision = "mx8" if cl & {"conv1", "conv2"} else "fp16"
        return n

    # ------------------------------------------------------------------ LoRA selection
    def set_lora_state(self, state: Optional[LoraState]) -> None:
        if self._linears is None:
            self._linears = [m for m in self.modules() if isinstance(m, Linear)]
        for m in self._linears:
            m.lora_state = state

    def refresh_cross_kv(self, ctx: torch.Tensor, state: Optional[LoraState]) -> None:
        """Recompute the cached cross-attention K / V^T of every attn2 for ``ctx`` (in place when the shape was seen
        before, so captured graphs stay valid).  Called eagerly before replaying step graphs on new inputs."""
        self.set_lora_state(state)
        try:
            for _, m in self.attentions():
                if m.is_cross:
                    m.project_cross(ctx)
        finally:
            self.set_lora_state(None)

    def refresh_ip_kv(self, ip_ctx: torch.Tensor) -> None:
        """Same for the image-prompt K / V^T of every attn2 that carries IP-Adapter weights (InstantID concept samples)."""
        from .attention import _ip_kv
        for _, m in self.attentions():
            if m.is_cross and getattr(m, "ip_kv_weight", None) is not None:
                _ip_kv(m, ip_ctx)

    # ------------------------------------------------------------------ forward
    def _boundary_weights(self):
        if not self._boundary:
            self._boundary["in"] = ops.pack_conv_in_weight(self.conv_in.weight.data)
            self._boundary["out"] = self.conv_out.weight.data.permute(0, 2, 3, 1).contiguous()
        return self._boundary

    @staticmethod
    def _nhwc(r: torch.Tensor, dt) -> torch.Tensor:
        """NCHW-shaped residual -> contiguous NHWC in `dt` (free when it already is a channels_last view, as the
        residuals of omg_amd.controlnet are)."""
        v = r.permute(0, 2, 3, 1)
        if v.dtype != dt:
            v = v.to(dt)
        return v if v.is_contiguous() else v.contiguous()

    def time_embed(self, timestep, B: int, text_embeds: torch.Tensor, time_ids: torch.Tensor) -> torch.Tensor:
        """emb = time_embedding(sincos(t)) + add_embedding([text_embeds | sincos(time_ids)])  -> (B, 4*C0)."""
        cfg = self.config
        dev, dt = self.device, self._dtype
        if not torch.is_tensor(timestep):
            t = torch.full((B,), float(timestep), dtype=torch.float32, device=dev)
        else:
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
            if t.numel() == 1:
                t = t.expand(B).contiguous()
        t_emb = ops.timestep_embedding(t.contiguous(), cfg.block_out_channels[0], dt)
        te_dim = text_embeds.shape[-1]
        add = torch.empty((B, cfg.projection_class_embeddings_input_dim), dtype=dt, device=dev)
        ops.copy2d(text_embeds.to(dt).contiguous(), add[:, :te_dim])
        tid = time_ids.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        n_ids = tid.numel() // B
        # time-id embeddings are written row by row into the tail of `add` (one row per id)
        tmp = ops.timestep_embedding(tid, cfg.addition_time_embed_dim, dt)              # (B*n_ids, dim)
        ops.copy2d(tmp.view(B, n_ids * cfg.addition_time_embed_dim), add[:, te_dim:])
        aug = self.add_embedding(add)
        return self.time_embedding(t_emb, residual=aug)

    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, class_labels=None,
                timestep_cond=None, attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals: Optional[Sequence[torch.Tensor]] = None,
                mid_block_additional_residual: Optional[torch.Tensor] = None, encoder_attention_mask=None,
                return_dict: bool = False, emb: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        """Returns ``(noise_pred,)`` (NCHW).  ``noise_pred`` is fp32 when ``sample`` is fp32 or ``out`` is
        given, else cast to ``sample.dtype`` like the reference.  ``emb`` may carry a precomputed
        time/text embedding (hoisted out of the step loop by omg_amd.pipeline)."""
        if attention_mask is not None or encoder_attention_mask is not None or class_labels is not None or timestep_cond is not None:
            raise L.OmgHipError("attention_mask / class_labels / timestep_cond are not used on OMG's SDXL path")
        if not sample.is_cuda:
            raise L.OmgHipError("UNet2DConditionModel runs on the MI355X only (no CPU fallback)")
        cfg = self.config
        B, _, H, W = sample.shape
        dt = self._dtype
        kw = dict(cross_attention_kwargs or {})
        kw.pop("scale", None)          # LoRA scale is baked into the LoRA bank (omg_amd.lora)
        row_residuals = kw.pop("omg_residuals", None)     # [(row0, row1, down[9], mid)]: ControlNet residuals for a row range
        if emb is None:
            if added_cond_kwargs is None:
                raise L.OmgHipError("added_cond_kwargs={'text_embeds','time_ids'} is required (addition_embed_type='text_time')")
            emb = self.time_embed(timestep, B, added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"])
        ctx = _Ctx(ops.silu(emb), encoder_hidden_states.to(dt).contiguous() if encoder_hidden_states.dtype != dt or not encoder_hidden_states.is_contiguous() else encoder_hidden_states, B)
        bw = self._boundary_weights()
        x_in = sample.contiguous()
        if x_in.dtype not in (torch.float32, dt):
            x_in = x_in.to(dt)
        h = ops.conv_in(x_in, bw["in"], self.conv_in.bias, dt)
        skips: List[torch.Tensor] = [h]
        for blk in self.down_blocks:
            h = blk(h, ctx, kw, skips)
        if down_block_additional_residuals is not None:
            if len(down_block_additional_residuals) != len(skips):
                raise ValueError(f"expected {len(skips)} down-block residuals, got {len(down_block_additional_residuals)}")
            skips[-1] = skips[-1].clone()                 # the last skip aliases `h`, which enters the mid block UN-modified
            for s_, r in zip(skips, down_block_additional_residuals):
                ops.add_(s_, self._nhwc(r, dt))           # `down_block_res_sample + residual` for every skip tensor
        if row_residuals:
            skips[-1] = skips[-1].clone() if down_block_additional_residuals is None else skips[-1]
            for r0, r1, down, _ in row_residuals:
                for s_, r in zip(skips, down):
                    ops.add_(s_[r0:r1], self._nhwc(r, dt))
        h = self.mid_block(h, ctx, kw)
        if mid_block_additional_residual is not None:
            ops.add_(h, self._nhwc(mid_block_additional_residual, dt))
        if row_residuals:
            for r0, r1, _, mid in row_residuals:
                ops.add_(h[r0:r1], self._nhwc(mid, dt))
        for blk in self.up_blocks:
            h = blk(h, ctx, kw, skips)
        h = self.conv_norm_out(h, silu=True)
        y = ops.conv_out(h, bw["out"], self.conv_out.bias, out=out)
        if out is None and sample.dtype != torch.float32:
            y = y.to(sample.dtype)
        if return_dict:
            return SimpleNamespace(sample=y)
        return (y,)
