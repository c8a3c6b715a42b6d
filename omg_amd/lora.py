"""Multi-adapter LoRA bank for the concept UNet.

Replaces PEFT's un-merged LoRA layers and ``concept_models.set_adapters(...)`` toggling
(/root/reference src/pipelines/lora_pipeline.py:340-342, :588-591; ``peft==0.8.2``,
``inference_lora.py:166-170``): ``y = base(x) + scale * w_a * (alpha_a / r_a) * B_a(A_a(x))`` summed over
the active adapters ``a`` (``[lora, "style"]`` with weights ``[0.7, 0.5]`` when ``styleL``).

MI355X-first: instead of walking the module tree in Python 68 times per image to switch adapters,
every *combination* that the pipeline will use becomes one "slot" — the active adapters' A matrices
stacked along the rank axis, their scaled B matrices stacked likewise — and all slots live in HBM at
once (``lora_down [slots, r, in]``, ``lora_up [slots, out, r]``).  A per-sample int32 vector then
selects the slot inside the GEMM (second K-segment, see ``omg_gemm``), so the K concept passes of a
step run as ONE batched forward with different adapters per sample, and the whole step stays
capturable in a hipGraph.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .modules import GEGLU, Linear


class LoraAdapter:
    """One named adapter: ``weights[module_path] = (A [r, in], B [out, r])`` and its ``alpha``."""

    def __init__(self, name: str, weights: Dict[str, Tuple[torch.Tensor, torch.Tensor]], alpha: Optional[float] = None):
        self.name, self.weights = name, weights
        ranks = {a.shape[0] for a, _ in weights.values()}
        if len(ranks) != 1:
            raise ValueError("per-layer ranks differ inside one adapter; not supported")
        self.rank = ranks.pop()
        self.alpha = float(alpha) if alpha is not None else float(self.rank)


def lora_target_names(unet) -> List[str]:
    """All attention + feed-forward Linear layers (synthetic coverage of SURVEY.md §8d)."""
    out = []
    for name, m in unet.named_modules():
        if isinstance(m, Linear) and any(s in name for s in (".to_q", ".to_k", ".to_v", ".to_out.0", ".ff.net.0.proj", ".ff.net.2")):
            out.append(name)
    return out


def make_synthetic_adapter(unet, name: str, rank: int, seed: int) -> LoraAdapter:
    """Random adapter generated on the device: A ~ N(0, 1/in), B ~ N(0, 1e-2)."""
    g = torch.Generator(device=unet.device).manual_seed(seed)
    w = {}
    for key in lora_target_names(unet):
        lin = unet.get_submodule(key)
        a = torch.randn(rank, lin.in_features, generator=g, device=unet.device) * lin.in_features ** -0.5
        b = torch.randn(lin.out_features, rank, generator=g, device=unet.device) * 0.1
        w[key] = (a.to(unet.dtype), b.to(unet.dtype))
    return LoraAdapter(name, w)


class LoraBank:
    def __init__(self, unet, adapters: Sequence[LoraAdapter]):
        self.unet = unet
        self.adapters = {a.name: a for a in adapters}
        self.slots: List[Tuple[Tuple[str, float], ...]] = []
        self.scale = 1.0
        self.mode = "merged"

    def slot_of(self, combo: Sequence[Tuple[str, float]]) -> int:
        return self.slots.index(tuple((n, float(w)) for n, w in combo))

    def build(self, slots: Sequence[Sequence[Tuple[str, float]]], scale: float = 1.0, mode: str = "merged") -> None:
        """Materialise the slot stacks on every target Linear.  ``slots[s]`` = [(adapter name, weight), ...];
        ``scale`` is ``cross_attention_kwargs['scale']`` (0.8 in OMG, lora_pipeline.py:596).

        mode="segment": keep A/B un-merged (PEFT's arithmetic: base(x) + s*B(A(x)), second K-segment of omg_gemm).
        mode="merged" : additionally build ``w_slots[1+S, out, in] = [W, W + s*B_1 A_1, ...]`` (fp32 merge, one
        rounding) so that a batch mixing base and concept samples runs ONE GEMM per layer with a per-sample weight
        slot and zero LoRA overhead — 4.4 GB of HBM per concept, which a 288 GB part has to spare."""
        if mode not in ("merged", "segment"):
            raise ValueError(mode)
        self.mode = mode
        self.slots = [tuple((n, float(w)) for n, w in s) for s in slots]
        self.scale = scale
        dev, dt = self.unet.device, self.unet.dtype
        keys = set()
        for a in self.adapters.values():
            keys.update(a.weights)
        r_tot = max(sum(self.adapters[n].rank for n, _ in s) for s in self.slots)
        r_tot = (r_tot + 7) // 8 * 8
        self.clear()
        for key in sorted(keys):
            lin = self.unet.get_submodule(key)
            if not isinstance(lin, Linear):
                raise L.OmgHipError(f"LoRA target {key} is not a Linear layer (conv LoRA is not supported)")
            down = torch.zeros(len(self.slots), r_tot, lin.in_features, dtype=torch.float32, device=dev)
            up = torch.zeros(len(self.slots), lin.out_features, r_tot, dtype=torch.float32, device=dev)
            for s, combo in enumerate(self.slots):
                r0 = 0
                for name, w in combo:
                    ad = self.adapters[name]
                    if key not in ad.weights:
                        r0 += ad.rank
                        continue
                    a, b = ad.weights[key]
                    down[s, r0:r0 + ad.rank] = a.to(dev).float()
                    up[s, :, r0:r0 + ad.rank] = b.to(dev).float() * (scale * w * ad.alpha / ad.rank)
                    r0 += ad.rank
            lin.lora_down = down.to(dt).contiguous()
            lin.lora_up = up.to(dt).contiguous()
            if mode == "merged":
                base = lin.weight.data.float()
                lin.w_slots = torch.stack([base] + [base + up[s_] @ down[s_] for s_ in range(len(self.slots))]).to(dt).contiguous()
        for m in self.unet.modules():
            if m is not self.unet and hasattr(m, "invalidate_packed") and not isinstance(m, Linear):
                m.invalidate_packed()

    def clear(self) -> None:
        for m in self.unet.modules():
            if isinstance(m, Linear):
                m.lora_down = m.lora_up = m.w_slots = None
            if m is not self.unet and hasattr(m, "invalidate_packed") and not isinstance(m, Linear):
                m.invalidate_packed()
