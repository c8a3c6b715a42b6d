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
from .attention import Attention
from .modules import GEGLU, Linear


class LoraAdapter:
    """One named adapter: ``weights[module_path] = (A [r, in], B [out, r])`` and its ``alpha``.

    Layers may have different ranks (PEFT ``rank_pattern``, common in kohya files): ``rank`` is the largest one — the width a
    layer occupies in the slot stacks, smaller layers are zero-padded — and the PEFT factor ``alpha / r`` uses each layer's own
    ``r``.  ``alpha=None`` means "alpha equals the rank of every layer" (factor 1), PEFT's and diffusers' default and what the
    file loaders produce after folding a per-layer ``alpha`` into B.  ``text_encoder`` optionally carries the text-encoder
    half of the file: ``{1 | 2: {module_path: (A, B)}}`` with factors already folded into B (omg_amd/loaders.py)."""

    def __init__(self, name: str, weights: Dict[str, Tuple[torch.Tensor, torch.Tensor]], alpha: Optional[float] = None,
                 text_encoder: Optional[Dict[int, Dict[str, Tuple[torch.Tensor, torch.Tensor]]]] = None):
        self.name, self.weights = name, weights
        self.rank = max(a.shape[0] for a, _ in weights.values()) if weights else 0
        self._alpha_given = alpha is not None
        self.alpha = float(alpha) if alpha is not None else float(self.rank)
        self.text_encoder = text_encoder or {}

    def scaling(self, key: str) -> float:
        """PEFT's ``lora_alpha / r`` of one layer."""
        return self.alpha / self.weights[key][0].shape[0] if self._alpha_given else 1.0


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
        self.version = 0             # advances with every build(): engines keyed on it never reuse stale slot stacks

    def slot_of(self, combo: Sequence[Tuple[str, float]]) -> int:
        return self.slots.index(tuple((n, float(w)) for n, w in combo))

    def build(self, slots: Sequence[Sequence[Tuple[str, float]]], scale: float = 1.0, mode: str = "merged") -> None:
        """Materialise the slot stacks on every target Linear.  ``slots[s]`` = [(adapter name, weight), ...];
        ``scale`` is ``cross_attention_kwargs['scale']`` (0.8 in OMG, lora_pipeline.py:596) — a float, or one value per slot.

        mode="segment": keep A/B un-merged (PEFT's arithmetic: base(x) + s*B(A(x)), second K-segment of omg_gemm).
        mode="merged" : additionally build ``w_slots[1+S, out, in] = [W, W + s*B_1 A_1, ...]`` (fp32 merge, one
        rounding) so that a batch mixing base and concept samples runs ONE GEMM per layer with a per-sample weight
        slot and zero LoRA overhead — 4.4 GB of HBM per concept, which a 288 GB part has to spare."""
        if mode not in ("merged", "segment"):
            raise ValueError(mode)
        self.mode = mode
        self.slots = [tuple((n, float(w)) for n, w in s) for s in slots]
        # one scale for every slot, or one per slot (the reference's concept passes hard-code 0.8, lora_pipeline.py:596, while
        # the main pass's style adapter takes the caller's cross_attention_kwargs["scale"], :546-566)
        per_slot = [float(x) for x in scale] if isinstance(scale, (list, tuple)) else [float(scale)] * len(self.slots)
        if len(per_slot) != len(self.slots):
            raise ValueError("one LoRA scale per slot")
        self.scale = tuple(per_slot) if isinstance(scale, (list, tuple)) else scale
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
                    r = a.shape[0]                                   # <= ad.rank; the rest of the adapter's band stays zero
                    down[s, r0:r0 + r] = a.to(dev).float()
                    up[s, :, r0:r0 + r] = b.to(dev).float() * (per_slot[s] * w * ad.scaling(key))
                    r0 += ad.rank
            lin.lora_down = down.to(dt).contiguous()
            lin.lora_up = up.to(dt).contiguous()
            if mode == "merged":
                base = lin.weight.data.float()
                lin.w_slots = torch.stack([base] + [base + up[s_] @ down[s_] for s_ in range(len(self.slots))]).to(dt).contiguous()
        if mode == "merged":
            # the fused q|k|v (k|v) GEMM takes one weight stack per slot: an attention whose adapter targets only some of
            # its projections (custom PEFT target_modules) gets the base weight repeated for the others
            for m in self.unet.modules():
                if isinstance(m, Attention):
                    projs = (m.to_q, m.to_k, m.to_v)
                    if any(l.w_slots is not None for l in projs):
                        for l in projs:
                            if l.w_slots is None:
                                l.w_slots = l.weight.data.unsqueeze(0).repeat(1 + len(self.slots), 1, 1).contiguous()
        self.version += 1
        self._invalidate_packed()

    def _invalidate_packed(self) -> None:
        """Only the modules whose packed images contain LoRA material (fused q|k|v stacks, GEGLU row-interleaved stacks):
        conv weights are untouched by a slot change and keep their packed form (and their pointers)."""
        for m in self.unet.modules():
            if isinstance(m, (GEGLU, Attention)):
                m.invalidate_packed()
            elif isinstance(m, Linear):
                m.invalidate_mx8_slots()

    def clear(self) -> None:
        for m in self.unet.modules():
            if isinstance(m, Linear):
                m.lora_down = m.lora_up = m.w_slots = None
        self._invalidate_packed()
