"""CLIP text transformers of SDXL's ``encode_prompt`` on the HIP kernels — row N4 of SURVEY.md §8f (the reference reaches them
through ``self.encode_prompt`` / ``concept_models.encode_prompt``, /root/reference src/pipelines/lora_pipeline.py:315-347).

State-dict keys equal transformers' ``CLIPTextModel`` / ``CLIPTextModelWithProjection`` (``text_model.embeddings.*``,
``text_model.encoder.layers.i.{self_attn.{q,k,v,out}_proj, layer_norm1, layer_norm2, mlp.fc1, mlp.fc2}``,
``text_model.final_layer_norm``, ``text_projection.weight``), so ``omg_amd.loaders.load_model_weights`` fills it from
``text_encoder/model.safetensors``.  Tokenisation stays with the caller (no tokenizer files exist offline).

This path runs once per prompt (77 tokens), so it is composed from kernels the UNet and VAE already validated rather than
given kernels of its own:
  * LayerNorm, q|k|v / out_proj / fc1 / fc2 as `omg_gemm` with bias and residual epilogues;
  * causal attention per (sample, head) = scores GEMM (out_scale 1/sqrt(64)) + additive mask (`omg_add_inplace`) +
    `omg_softmax_rows` + P·V GEMM — the flash kernel has no causal mask yet;
  * quick_gelu(x) = silu(1.702 x) / 1.702: fc1 with ``out_scale`` 1.702, `omg_silu`, and fc2's weights divided by 1.702;
  * exact gelu (OpenCLIP-bigG): fc1 run as a GEGLU GEMM whose value half is the constant 1 (zero weights, bias 1).
Sequences are padded from 77 to 80 positions (GEMM extents are multiples of 8); the causal mask keeps the padding out of
every real position.  The embedding lookup is a torch gather (data movement, no arithmetic).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from . import _lib as L
from . import ops
from .modules import LayerNorm, Linear


class ClipTextConfig:
    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=768, eos_token_id=49407,
                 with_projection=True):
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.max_position_embeddings, self.hidden_act, self.layer_norm_eps = max_position_embeddings, hidden_act, layer_norm_eps
        self.projection_dim, self.eos_token_id, self.with_projection = projection_dim, eos_token_id, with_projection
        if hidden_size != 64 * num_attention_heads:
            raise ValueError("head_dim must be 64 (both SDXL text encoders: 768/12, 1280/20)")
        if hidden_act not in ("quick_gelu", "gelu"):
            raise ValueError(hidden_act)

    @staticmethod
    def clip_l() -> "ClipTextConfig":
        return ClipTextConfig(with_projection=False)

    @staticmethod
    def open_clip_bigg() -> "ClipTextConfig":
        return ClipTextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                              hidden_act="gelu", projection_dim=1280)


class _Layer(nn.Module):
    def __init__(self, cfg, dtype, device):
        super().__init__()
        d, f = cfg.hidden_size, cfg.intermediate_size
        self.self_attn = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self.self_attn, n, Linear(d, d, dtype=dtype, device=device))
        self.layer_norm1 = LayerNorm(d, cfg.layer_norm_eps, dtype=dtype, device=device)
        self.layer_norm2 = LayerNorm(d, cfg.layer_norm_eps, dtype=dtype, device=device)
        self.mlp = nn.Module()
        self.mlp.fc1 = Linear(d, f, dtype=dtype, device=device)
        self.mlp.fc2 = Linear(f, d, dtype=dtype, device=device)


class ClipTextEncoder(nn.Module):
    TP = 80          # padded sequence length

    def __init__(self, cfg: ClipTextConfig, dtype=torch.float16, device="cuda"):
        super().__init__()
        self.config, self._dtype = cfg, dtype
        d = cfg.hidden_size
        self.text_model = nn.Module()
        emb = nn.Module()
        emb.token_embedding = nn.Embedding(cfg.vocab_size, d, dtype=dtype, device=device)
        emb.position_embedding = nn.Embedding(cfg.max_position_embeddings, d, dtype=dtype, device=device)
        self.text_model.embeddings = emb
        self.text_model.encoder = nn.Module()
        self.text_model.encoder.layers = nn.ModuleList([_Layer(cfg, dtype, device) for _ in range(cfg.num_hidden_layers)])
        self.text_model.final_layer_norm = LayerNorm(d, cfg.layer_norm_eps, dtype=dtype, device=device)
        if cfg.with_projection:
            self.text_projection = nn.Module()
            self.text_projection.weight = nn.Parameter(torch.empty(cfg.projection_dim, d, dtype=dtype, device=device), requires_grad=False)
        for p in self.parameters():
            p.requires_grad_(False)
        self._packed = None

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def set_lora(self, combo=None) -> None:
        """Activate text-encoder LoRA deltas for the following forwards: ``combo`` = [(weights, multiplier), ...] with
        ``weights[module path] = (A [r, in], B [out, r])`` (``LoraAdapter.text_encoder[n]``) and multiplier =
        ``lora_scale * adapter_weight`` — PEFT's ``set_adapters`` + ``scale_lora_layers(text_encoder, lora_scale)`` as
        ``encode_prompt`` runs them (lora_pipeline.py:336-347).  The deltas are merged into the packed weight images in fp32
        (one rounding); ``None`` restores the base weights."""
        combo = [(w, float(m)) for w, m in (combo or []) if w]
        sig = tuple((id(w), m) for w, m in combo)
        if sig != getattr(self, "_lora_sig", ()):
            self._lora, self._lora_sig, self._packed = combo, sig, None

    def _w(self, li: int, sub: str, lin: Linear) -> torch.Tensor:
        """Weight of ``text_model.encoder.layers.<li>.<sub>`` with the active LoRA deltas merged in."""
        w = lin.weight.data
        key = f"text_model.encoder.layers.{li}.{sub}"
        hits = [(wd[key], m) for wd, m in getattr(self, "_lora", None) or [] if key in wd]
        if not hits:
            return w
        acc = w.float()
        for (a, b), m in hits:
            acc = acc + m * (b.to(device=w.device, dtype=torch.float32) @ a.to(device=w.device, dtype=torch.float32))
        return acc.to(w.dtype)

    def _pack(self):
        """Derived weight images: q|k|v rows concatenated; quick_gelu's 1/1.702 folded into fc2; gelu's fc1 as a GEGLU GEMM with
        a constant-one value half; the causal mask; active LoRA deltas merged."""
        cfg, dev, dt = self.config, self.device, self._dtype
        layers = []
        for li, lyr in enumerate(self.text_model.encoder.layers):
            a = lyr.self_attn
            pk = {"wqkv": torch.cat([self._w(li, "self_attn.q_proj", a.q_proj), self._w(li, "self_attn.k_proj", a.k_proj),
                                     self._w(li, "self_attn.v_proj", a.v_proj)]).contiguous(),
                  "bqkv": torch.cat([a.q_proj.bias.data, a.k_proj.bias.data, a.v_proj.bias.data]).contiguous(),
                  "wo": self._w(li, "self_attn.out_proj", a.out_proj)}
            w1, w2 = self._w(li, "mlp.fc1", lyr.mlp.fc1), self._w(li, "mlp.fc2", lyr.mlp.fc2)
            if cfg.hidden_act == "quick_gelu":
                pk["w1"] = w1
                pk["w2"] = (w2.float() / 1.702).to(dt).contiguous()
            else:
                f = cfg.intermediate_size
                perm = ops.geglu_row_perm(2 * f).to(dev)
                w1g = torch.cat([torch.zeros_like(w1), w1])
                b1 = torch.cat([torch.ones_like(lyr.mlp.fc1.bias.data), lyr.mlp.fc1.bias.data])
                pk["w1g"], pk["b1g"] = w1g[perm].contiguous(), b1[perm].contiguous()
                pk["w2"] = w2
            layers.append(pk)
        T = self.TP
        mask = torch.zeros(T, T, dtype=dt, device=dev)
        mask.masked_fill_(torch.ones(T, T, dtype=torch.bool, device=dev).triu(1), -30000.0)
        pos = torch.zeros(T, cfg.hidden_size, dtype=dt, device=dev)
        pos[: cfg.max_position_embeddings] = self.text_model.embeddings.position_embedding.weight.data
        self._packed = {"layers": layers, "mask": mask, "pos": pos}

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, clip_skip: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """``input_ids`` (B, 77) int64 on the device -> (hidden_states[-2] (B, 77, d), pooled (B, projection_dim or d)),
        i.e. what diffusers' encode_prompt takes from ``text_encoder(ids, output_hidden_states=True)``.  ``clip_skip=k`` (lora_pipeline.py:245, :333):
        ``hidden_states[-(k + 2)]`` instead — SDXL counts from the penultimate layer —; the pooled output is the full pass's either way."""
        cfg = self.config
        if not input_ids.is_cuda:
            raise L.OmgHipError("ClipTextEncoder runs on the HIP kernels only (no CPU fallback)")
        if self._packed is None:
            self._pack()
        pk = self._packed
        B, T0 = input_ids.shape
        T, d, heads = self.TP, cfg.hidden_size, cfg.num_attention_heads
        if T0 > cfg.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        ids = torch.full((B, T), cfg.eos_token_id, dtype=torch.long, device=input_ids.device)
        ids[:, :T0] = input_ids
        x = (self.text_model.embeddings.token_embedding.weight.data[ids] + pk["pos"]).reshape(B * T, d).contiguous()
        scores = torch.empty(T, T, dtype=x.dtype, device=x.device)
        penultimate = x
        n_layers = len(self.text_model.encoder.layers)
        skip = int(clip_skip or 0)
        if not 0 <= skip < n_layers:
            raise ValueError(f"clip_skip={clip_skip}: this encoder has {n_layers} layers")
        for li, (lyr, w) in enumerate(zip(self.text_model.encoder.layers, pk["layers"])):
            if li == n_layers - 1 - skip:
                penultimate = x                                     # hidden_states[-(2 + skip)] = the input of layer n - 1 - skip
            h = lyr.layer_norm1(x)
            qkv = ops.gemm(h, w["wqkv"], bias=w["bqkv"]).view(B, T, 3 * d)
            vt = ops.transpose_v(qkv[:, :, 2 * d:], heads, mfma_order=False)   # (B, heads, 64, 128), plain transpose
            attn = torch.empty(B * T, d, dtype=x.dtype, device=x.device)
            for b in range(B):
                for hh in range(heads):
                    ops.gemm(qkv[b, :, hh * 64:(hh + 1) * 64], qkv[b, :, d + hh * 64: d + (hh + 1) * 64], out=scores, out_scale=0.125)
                    ops.add_(scores, pk["mask"])
                    ops.softmax_rows_(scores, 1.0)
                    ops.gemm(scores, vt[b, hh, :, :T], out=attn[b * T:(b + 1) * T, hh * 64:(hh + 1) * 64])
            x = ops.gemm(attn, w["wo"], bias=lyr.self_attn.out_proj.bias.data, residual=x)
            h = lyr.layer_norm2(x)
            if cfg.hidden_act == "quick_gelu":
                f = ops.silu(ops.gemm(h, w["w1"], bias=lyr.mlp.fc1.bias.data, out_scale=1.702))
            else:
                f = ops.gemm(h, w["w1g"], bias=w["b1g"], act=L.ACT_GEGLU)
            x = ops.gemm(f, w["w2"], bias=lyr.mlp.fc2.bias.data, residual=x)
        last = self.text_model.final_layer_norm(x).view(B, T, d)
        eos = (ids == cfg.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(B, device=ids.device), eos].contiguous()
        if cfg.with_projection:
            pooled = ops.gemm(pooled, self.text_projection.weight.data)
        return penultimate.view(B, T, d)[:, :T0].contiguous(), pooled


@torch.no_grad()
def encode_prompt(enc_l: ClipTextEncoder, enc_g: ClipTextEncoder, ids_l: torch.Tensor, ids_g: torch.Tensor,
                  clip_skip: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """SDXL's two-encoder embedding: (B, 77, 2048) and the pooled (B, 1280) of the second encoder."""
    hl, _ = enc_l(ids_l, clip_skip)
    hg, pooled = enc_g(ids_g, clip_skip)
    return torch.cat([hl, hg], dim=-1), pooled


def make_encode_prompt(enc_l: ClipTextEncoder, enc_g: ClipTextEncoder, tokenize_l, tokenize_g=None, adapters=None):
    """The ``encode_prompt=`` callable of :class:`omg_amd.pipeline.LoraMultiConceptPipeline`:
    ``fn(prompt, negative_prompt, lora_param, lora_scale) -> (prompt_embeds, negative_prompt_embeds, pooled, negative_pooled)`` as
    diffusers' ``StableDiffusionXLPipeline.encode_prompt`` returns them (lora_pipeline.py:315-347).  ``tokenize_*`` map a list of
    strings to (B, 77) int64 ids (the reference's ``pipe.tokenizer`` / ``pipe.tokenizer_2`` with ``padding="max_length"``).  A
    ``None`` negative prompt gives zero embeddings (SDXL-base ships ``force_zeros_for_empty_prompt=True``).

    Round 6: ``prompt_2`` / ``negative_prompt_2`` (lora_pipeline.py:215, :222: the SECOND encoder's own prompt, default = the first's) and
    ``clip_skip`` (:245, :333) as diffusers' ``encode_prompt`` takes them; like there, ``clip_skip`` moves the POSITIVE prompt's hidden state
    only — the negative prompt is always read at ``hidden_states[-2]`` (diffusers 0.25.0, recalled).

    ``lora_param``: the adapters active on the text encoders while this prompt is encoded — ``None``, an adapter name, or
    [(name, weight), ...] (the reference calls ``concept_models.set_adapters(lora)`` / ``set_adapters([lora, "style"], [0.7, 0.5])``
    right before ``concept_models.encode_prompt(..., lora_scale=0.8)``, lora_pipeline.py:336-347).  ``adapters`` maps names to
    :class:`omg_amd.lora.LoraAdapter` objects whose ``text_encoder[1]`` / ``[2]`` hold the CLIP-L / OpenCLIP-bigG halves of the LoRA
    file (omg_amd/loaders.py); an adapter without text-encoder entries changes nothing.  When ``adapters`` is given, a name missing from
    it is an error; when it is omitted (``make_encode_prompt(enc_l, enc_g, tok_l, tok_g)``), every adapter is taken to be UNet-only."""
    tokenize_g = tokenize_g or tokenize_l
    strict = adapters is not None      # an explicit table must know every name; without one, adapters have no text-encoder half
    adapters = adapters if adapters is not None else {}

    def _list(p, n):
        out = [p] if isinstance(p, str) else list(p)
        return out * n if len(out) == 1 and n > 1 else out

    def fn(prompt, negative_prompt=None, lora_param=None, lora_scale=None, prompt_2=None, negative_prompt_2=None, clip_skip=None):
        combo = [] if lora_param is None else [(lora_param, 1.0)] if isinstance(lora_param, str) else list(lora_param)
        ls = 1.0 if lora_scale is None else float(lora_scale)
        for n_te, enc in ((1, enc_l), (2, enc_g)):
            act = []
            for name, w in combo:
                if name not in adapters:
                    if strict:
                        raise KeyError(f"text-encoder LoRA: adapter {name!r} was not given to make_encode_prompt(adapters=...)")
                    continue       # a UNet-only adapter changes nothing in the text encoders (PEFT: no lora layers there)
                act.append((adapters[name].text_encoder.get(n_te), ls * float(w)))
            enc.set_lora(act)
        try:
            prompts = _list(prompt, 1)
            prompts_2 = _list(prompt_2, len(prompts)) if prompt_2 is not None else prompts
            if len(prompts_2) != len(prompts):
                raise ValueError("prompt_2 must be one string or one per prompt")
            dev = enc_l.device
            pe, pp = encode_prompt(enc_l, enc_g, tokenize_l(prompts).to(dev), tokenize_g(prompts_2).to(dev), clip_skip)
            if negative_prompt is None and negative_prompt_2 is None:
                return pe, torch.zeros_like(pe), pp, torch.zeros_like(pp)
            negs = _list(negative_prompt if negative_prompt is not None else "", len(prompts))
            negs_2 = _list(negative_prompt_2, len(prompts)) if negative_prompt_2 is not None else negs
            if len(negs) != len(prompts) or len(negs_2) != len(prompts):
                raise ValueError("negative_prompt / negative_prompt_2 must be one string or one per prompt")
            ne, npp = encode_prompt(enc_l, enc_g, tokenize_l(negs).to(dev), tokenize_g(negs_2).to(dev))
            return pe, ne, pp, npp
        finally:
            enc_l.set_lora(None)
            enc_g.set_lora(None)

    fn.adapters = adapters
    return fn
