"""Synthetic inputs of the benchmark configurations (SURVEY.md §8d recipe).  No SDXL weights, LoRA files,
text encoders or datasets exist offline, so the bench uses seeded random weights of the exact SDXL-base
architecture and random embeddings of the exact shapes; results are labelled ``"data": "synthetic"``."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .lora import LoraBank, make_synthetic_adapter
from .pipeline import ConceptModels


def c2_masks(height: int = 1024, width: int = 1024, device="cpu") -> List[torch.Tensor]:
    """Config 2 masks: concept 1 = rows 256.. x cols 64..479, concept 2 = rows 256.. x cols 448..959
    (32-px overlap on purpose: exercises the sum rule of lora_pipeline.py:602)."""
    s = height / 1024.0
    m1 = torch.zeros(height, width, device=device)
    m2 = torch.zeros(height, width, device=device)
    m1[int(256 * s):, int(64 * s): int(480 * s)] = 1
    m2[int(256 * s):, int(448 * s): int(960 * s)] = 1
    return [m1, m2]


def c2_inputs(unet, seed: int, n_concepts: int = 2, height: int = 1024, width: int = 1024) -> Dict:
    """Random prompt/pooled embeddings for the global prompt [p, p] and each region prompt, and the seed-14 latents."""
    cfg = unet.config
    dev, dt = unet.device, unet.dtype
    g = torch.Generator(device="cpu").manual_seed(seed)
    cx = cfg.cross_attention_dim
    pooled = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim

    def emb(n):
        return torch.randn(n, 77, cx, generator=g).to(dt).to(dev), torch.randn(n, pooled, generator=g).to(dt).to(dev)

    pe, pp = emb(1)
    ne, npp = emb(1)
    regions = []
    for _ in range(n_concepts):
        re_, rp_ = emb(2)
        regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))      # (neg_embeds, pos_embeds, neg_pooled, pos_pooled)
    lat = torch.randn(1, cfg.in_channels, height // 8, width // 8, generator=torch.Generator().manual_seed(14 + seed))
    return dict(prompt_embeds=pe.repeat(2, 1, 1), negative_prompt_embeds=ne.repeat(2, 1, 1), pooled_prompt_embeds=pp.repeat(2, 1),
                negative_pooled_prompt_embeds=npp.repeat(2, 1), region_prompt_embeds=regions, latents=lat)


def make_concept_models(unet, n_concepts: int = 2, rank: int = 64, style: bool = False) -> ConceptModels:
    adapters = [make_synthetic_adapter(unet, f"concept{c}", rank, seed=1000 + c) for c in range(n_concepts)]
    if style:
        adapters.append(make_synthetic_adapter(unet, "style", rank, seed=1999))
    return ConceptModels(unet, LoraBank(unet, adapters))
