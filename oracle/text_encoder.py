"""ORACLE (test infrastructure): fp32 restatement of the two CLIP text transformers behind SDXL's ``encode_prompt`` — row N4
of SURVEY.md §8f (the caller side of the hot path; /root/reference src/pipelines/lora_pipeline.py:315-347 calls
``self.encode_prompt`` / ``concept_models.encode_prompt``, i.e. diffusers' ``StableDiffusionXLPipeline.encode_prompt``):

    for each of (CLIP-L: 12 layers, width 768, 12 heads, quick_gelu | OpenCLIP-bigG: 32 layers, width 1280, 20 heads, gelu):
        out = text_encoder(input_ids, output_hidden_states=True)
        hidden = out.hidden_states[-2]                   # penultimate layer, NOT final-layer-normed
        pooled = out[0]                                  # (second encoder only) projected embedding of the EOS token
    prompt_embeds = concat(hidden_L, hidden_bigG, dim=-1)   # (B, 77, 768 + 1280 = 2048)

Third-party arithmetic (``transformers`` ``CLIPTextModel`` / ``CLIPTextModelWithProjection``): pre-LayerNorm transformer with a
causal mask, learned position embeddings, ``final_layer_norm`` before pooling, ``text_projection`` without bias.
PARITY PINNED: ``transformers`` IS importable here, so tests/test_oracle_text.py checks this restatement against
``CLIPTextModelWithProjection`` on seeded random weights for both activation variants (max |d| < 1e-5).
The HIP product for this row does not exist yet (it needs a causal flag in the attention kernel and gelu / quick_gelu GEMM
epilogues); this file is the pinned checker it will be built against.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class ClipTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    projection_dim: int = 768
    eos_token_id: int = 49407

    @staticmethod
    def clip_l() -> "ClipTextConfig":
        return ClipTextConfig()

    @staticmethod
    def open_clip_bigg() -> "ClipTextConfig":
        return ClipTextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                              hidden_act="gelu", projection_dim=1280)


def param_shapes(cfg: ClipTextConfig, with_projection: bool = True) -> Dict[str, Tuple[int, ...]]:
    d, f = cfg.hidden_size, cfg.intermediate_size
    out = {"text_model.embeddings.token_embedding.weight": (cfg.vocab_size, d),
           "text_model.embeddings.position_embedding.weight": (cfg.max_position_embeddings, d)}
    for i in range(cfg.num_hidden_layers):
        p = f"text_model.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[f"{p}.self_attn.{n}.weight"], out[f"{p}.self_attn.{n}.bias"] = (d, d), (d,)
        for n in ("layer_norm1", "layer_norm2"):
            out[f"{p}.{n}.weight"], out[f"{p}.{n}.bias"] = (d,), (d,)
        out[f"{p}.mlp.fc1.weight"], out[f"{p}.mlp.fc1.bias"] = (f, d), (f,)
        out[f"{p}.mlp.fc2.weight"], out[f"{p}.mlp.fc2.bias"] = (d, f), (d,)
    out["text_model.final_layer_norm.weight"], out["text_model.final_layer_norm.bias"] = (d,), (d,)
    if with_projection:
        out["text_projection.weight"] = (cfg.projection_dim, d)
    return out


def _act(x: Tensor, name: str) -> Tensor:
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return F.gelu(x)
    raise ValueError(name)


def _ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"].float(), sd[name + ".bias"].float(), eps)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"].float(), sd[name + ".bias"].float())


def text_model(sd: Dict[str, Tensor], cfg: ClipTextConfig, input_ids: Tensor) -> Tuple[List[Tensor], Tensor, Tensor]:
    """Returns (hidden_states [embedding output, layer 1, ..., layer N], last_hidden_state after final_layer_norm,
    pooled = text_projection(last_hidden_state at the first EOS position) or the un-projected vector if no projection)."""
    B, T = input_ids.shape
    h = sd["text_model.embeddings.token_embedding.weight"].float()[input_ids] + \
        sd["text_model.embeddings.position_embedding.weight"].float()[:T]
    heads, d = cfg.num_attention_heads, cfg.hidden_size
    hd = d // heads
    causal = torch.full((T, T), float("-inf")).triu(1)
    hidden = [h]
    for i in range(cfg.num_hidden_layers):
        p = f"text_model.encoder.layers.{i}"
        x = _ln(sd, p + ".layer_norm1", h, cfg.layer_norm_eps)
        q = _lin(sd, p + ".self_attn.q_proj", x).view(B, T, heads, hd).transpose(1, 2)
        k = _lin(sd, p + ".self_attn.k_proj", x).view(B, T, heads, hd).transpose(1, 2)
        v = _lin(sd, p + ".self_attn.v_proj", x).view(B, T, heads, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5 + causal, dim=-1) @ v
        h = h + _lin(sd, p + ".self_attn.out_proj", a.transpose(1, 2).reshape(B, T, d))
        x = _ln(sd, p + ".layer_norm2", h, cfg.layer_norm_eps)
        h = h + _lin(sd, p + ".mlp.fc2", _act(_lin(sd, p + ".mlp.fc1", x), cfg.hidden_act))
        hidden.append(h)
    last = _ln(sd, "text_model.final_layer_norm", h, cfg.layer_norm_eps)
    eos = (input_ids == cfg.eos_token_id).int().argmax(dim=-1)
    pooled = last[torch.arange(B), eos]
    if "text_projection.weight" in sd:
        pooled = pooled @ sd["text_projection.weight"].float().t()
    return hidden, last, pooled


def encode_prompt(sd_l, cfg_l, sd_g, cfg_g, ids_l: Tensor, ids_g: Tensor) -> Tuple[Tensor, Tensor]:
    """SDXL's two-encoder prompt embedding: (B, 77, 2048) penultimate hidden states, (B, 1280) pooled (second encoder)."""
    hl, _, _ = text_model(sd_l, cfg_l, ids_l)
    hg, _, pooled = text_model(sd_g, cfg_g, ids_g)
    return torch.cat([hl[-2], hg[-2]], dim=-1), pooled
