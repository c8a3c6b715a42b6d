"""IP-Adapter (InstantID) weights for the concept UNet — rows A12 / config 3 of SURVEY.md §8.

Mirrors ``StableDiffusionXLInstantIDPipeline.set_ip_adapter`` (/root/reference src/pipelines/instantid_single_pieline.py:186-213):
every cross-attention (attn2) of the UNet gets ``to_k_ip`` / ``to_v_ip`` Linear(cross_attention_dim -> hidden) and a scale;
the checkpoint's ``ip_adapter`` state dict is keyed by the layer's index in ``unet.attn_processors`` order
(``"{idx}.to_k_ip.weight"``), exactly as ``torch.nn.ModuleList(unet.attn_processors.values()).load_state_dict`` expects.

The arithmetic (src/ip_adapter/attention_processor.py:383-409) — ``SDPA(q, K_text, V_text) + scale * SDPA(q, K_ip, V_ip)`` — runs as
a second, accumulating call of the flash kernel over the 16 image-prompt tokens (``omg_attn_fwd`` ``accumulate``).  Because the
concept samples share the batched forward with the main samples (which have no ip tokens), the weights hang on the ``Attention``
modules and the processors apply them to the row range named by ``omg_ip_rows``.
"""
from __future__ import annotations

from typing import Dict

import torch

from .attention import Attention
from .modules import bump_pointer_epoch


def attn_processor_index(unet) -> Dict[str, int]:
    """Attention module name -> its position in diffusers' ``unet.attn_processors`` (attn1 and attn2 both count).

    diffusers' ``UNet2DConditionModel.__init__`` creates ``down_blocks`` and ``up_blocks`` before ``mid_block``, so the
    processors enumerate down (SDXL: 0-47), up (48-119), mid (120-139) [recalled: diffusers is not installed here]; the
    ``ip_adapter`` checkpoint is the state dict of ``ModuleList(unet.attn_processors.values())``
    (instantid_single_pieline.py:208-212), i.e. keyed by exactly this index.  Computed from the block prefixes, not from
    this package's own module registration order."""
    rank = {"down_blocks": 0, "up_blocks": 1, "mid_block": 2}
    names = [name for name, _ in unet.attentions()]
    order = sorted(range(len(names)), key=lambda i: (rank[names[i].split(".")[0]], i))
    return {names[i]: pos for pos, i in enumerate(order)}


class IPAdapter:
    def __init__(self, unet, num_tokens: int = 16, scale: float = 0.5):
        self.unet, self.num_tokens, self.scale = unet, num_tokens, scale
        self.layers = []            # (index in diffusers' attn_processors order, module name, Attention)
        index = attn_processor_index(unet)
        for name, m in unet.attentions():
            if m.is_cross:
                self.layers.append((index[name], name, m))
        self.layers.sort(key=lambda t: t[0])

    def _install(self, m: Attention, wk: torch.Tensor, wv: torch.Tensor) -> None:
        dev, dt = self.unet.device, self.unet.dtype
        want = (m.inner_dim, m.to_k.in_features)
        if tuple(wk.shape) != want or tuple(wv.shape) != want:
            raise ValueError(f"ip-adapter weights {tuple(wk.shape)} / {tuple(wv.shape)} do not fit this attention layer "
                             f"(to_k_ip / to_v_ip must be {want}): wrong layer order or wrong checkpoint")
        m.ip_kv_weight = torch.cat([wk, wv], dim=0).to(device=dev, dtype=dt).contiguous()      # [2C, Cx]
        m.ip_scale, m.ip_tokens = self.scale, self.num_tokens
        m._ip_cache = None
        bump_pointer_epoch()        # captured step graphs hold the old weight / cache pointers

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        if "ip_adapter" in sd:
            sd = sd["ip_adapter"]
        for idx, name, m in self.layers:
            self._install(m, sd[f"{idx}.to_k_ip.weight"], sd[f"{idx}.to_v_ip.weight"])

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The ``ip_adapter`` half of an InstantID ``ip-adapter.bin`` for the installed weights (tests, export)."""
        out = {}
        for idx, name, m in self.layers:
            c = m.inner_dim
            out[f"{idx}.to_k_ip.weight"], out[f"{idx}.to_v_ip.weight"] = m.ip_kv_weight[:c].clone(), m.ip_kv_weight[c:].clone()
        return out

    def load_named(self, weights: Dict[str, tuple]) -> None:
        """weights[attn2 module name] = (to_k_ip, to_v_ip)."""
        for idx, name, m in self.layers:
            self._install(m, *weights[name])

    def init_synthetic_(self, seed: int = 0) -> "IPAdapter":
        g = torch.Generator(device=self.unet.device).manual_seed(seed)
        cx = self.unet.config.cross_attention_dim
        for idx, name, m in self.layers:
            c = m.inner_dim
            wk = torch.randn(c, cx, generator=g, device=self.unet.device) * cx ** -0.5
            wv = torch.randn(c, cx, generator=g, device=self.unet.device) * cx ** -0.5
            self._install(m, wk, wv)
        return self

    def set_scale(self, scale: float) -> None:
        if scale != self.scale:
            bump_pointer_epoch()    # the scale is a launch argument of the captured attention kernels
        self.scale = scale
        for _, _, m in self.layers:
            m.ip_scale = scale

    def remove(self) -> None:
        bump_pointer_epoch()
        for _, _, m in self.layers:
            m.ip_kv_weight = None
