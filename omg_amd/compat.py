"""The diffusers-shaped constructors and methods that the reference's scripts call — boundary B3 of SURVEY.md §8b.

``inference_lora.py`` builds its models with (``build_model_sd``, :152-171)

    controlnet   = ControlNetModel.from_pretrained(path, torch_dtype=torch.float16).to(device)
    pipe         = LoraMultiConceptPipeline.from_pretrained(model, controlnet=controlnet, torch_dtype=torch.float16, variant="fp16").to(device)
    controller   = AttentionReplace(prompts, 50, ..., tokenizer=pipe.tokenizer, device=device, dtype=torch.float16, width=, height=)
    revise_regionally_controlnet_forward(pipe.unet, controller)
    pipe_concept = StableDiffusionXLPipeline.from_pretrained(model, torch_dtype=torch.float16, variant="fp16").to(device)
    pipe_concept.enable_xformers_memory_efficient_attention()
    pipe.load_lora_weights(style, weight_name="pytorch_lora_weights.safetensors", adapter_name="style")
    pipe_concept.load_lora_weights(path, weight_name="pytorch_lora_weights.safetensors", adapter_name=name)

and then calls ``pipe(prompt=[[p, p], [(rp, rn), ...]], negative_prompt=[n, n], generator=, image=None | [pil, pil], ...,
concept_models=pipe_concept, ...).images`` (``sample_image``, :37-73) and saves ``images[0]`` / ``images[1]`` as PIL images.
This module provides those names on top of :mod:`omg_amd`, so that the edit to the reference's scripts is the import block only
(INTEGRATION.md §1).  Model directories are diffusers' on-disk layout (``model_index.json``; ``unet/``, ``vae/``,
``text_encoder/``, ``text_encoder_2/`` each with ``config.json`` + ``*.safetensors``; ``tokenizer/``, ``tokenizer_2/``;
``scheduler/scheduler_config.json``); tokenisation is transformers' ``CLIPTokenizer`` on the directory's own files.

MI355X-first difference, value-preserving: the reference holds TWO copies of the SDXL weights (main pipe + concept pipe, ~14 GB);
here ``from_pretrained`` of the same directory / dtype / variant returns pipelines that SHARE one UNet, one VAE decoder and one
pair of text encoders — the concept pipe differs from the main pipe only by its LoRA adapters, which live in per-sample weight
slots of a common :class:`omg_amd.lora.LoraBank`.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import loaders
from .controlnet import ControlNetModel as _ControlNetModel
from .ip_adapter import IPAdapter
from .lora import LoraBank
from .pipeline import (ConceptModels, InstantidMultiConceptPipeline as _InstantidPipe, LoraMultiConceptPipeline as _LoraPipe,
                       StableDiffusionXLPipelineOutput)
from .resampler import Resampler
from .schedulers import DDIMScheduler, EulerDiscreteScheduler
from .text_encoder import ClipTextConfig, ClipTextEncoder, make_encode_prompt
from .unet import UNet2DConditionModel, UNetConfig
from .vae import AutoencoderKLDecoder, VaeConfig


# ------------------------------------------------------------------------------------------------ files
def _json(path: str) -> dict:
    with open(path) as f:
        return json.load(f)


def _weights_file(folder: str, stem: str, variant: Optional[str]) -> str:
    """diffusers' naming: ``<stem>[.<variant>].safetensors`` (``.bin`` as a fallback)."""
    cands = []
    if variant:
        cands += [f"{stem}.{variant}.safetensors", f"{stem}.{variant}.bin"]
    cands += [f"{stem}.safetensors", f"{stem}.bin"]
    for c in cands:
        p = os.path.join(folder, c)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"no weights in {folder} (looked for {cands})")


def unet_config_from_dict(c: dict) -> UNetConfig:
    """diffusers ``unet/config.json`` -> :class:`UNetConfig`; anything outside the SDXL family is an error, not a silent default."""
    tl = c.get("transformer_layers_per_block", 1)
    n = len(c["block_out_channels"])
    ahd = c.get("attention_head_dim", 8)
    if c.get("addition_embed_type", "text_time") != "text_time" or not c.get("use_linear_projection", True):
        raise L.OmgHipError("only SDXL-family UNets (addition_embed_type='text_time', use_linear_projection=True) are supported")
    return UNetConfig(in_channels=c.get("in_channels", 4), out_channels=c.get("out_channels", 4), sample_size=c.get("sample_size", 128),
                      block_out_channels=tuple(c["block_out_channels"]), down_block_types=tuple(c["down_block_types"]),
                      up_block_types=tuple(c.get("up_block_types", ())), layers_per_block=c.get("layers_per_block", 2),
                      transformer_layers_per_block=tuple(tl) if isinstance(tl, (list, tuple)) else (tl,) * n,
                      attention_head_dim=tuple(ahd) if isinstance(ahd, (list, tuple)) else (ahd,) * n,
                      cross_attention_dim=c.get("cross_attention_dim", 2048), addition_time_embed_dim=c.get("addition_time_embed_dim", 256),
                      projection_class_embeddings_input_dim=c.get("projection_class_embeddings_input_dim", 2816),
                      norm_num_groups=c.get("norm_num_groups", 32), norm_eps=c.get("norm_eps", 1e-5))


def _clip_config(c: dict, with_projection: bool) -> ClipTextConfig:
    return ClipTextConfig(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                          num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                          max_position_embeddings=c.get("max_position_embeddings", 77), hidden_act=c.get("hidden_act", "quick_gelu"),
                          layer_norm_eps=c.get("layer_norm_eps", 1e-5), projection_dim=c.get("projection_dim", c["hidden_size"]),
                          eos_token_id=c.get("eos_token_id", 49407), with_projection=with_projection)


class _Components:
    """Everything one model directory provides, loaded once per (path, dtype, variant) and shared by the pipelines built from it."""

    def __init__(self, path: str, dtype: torch.dtype, variant: Optional[str]):
        self.path, self.dtype, self.variant = path, dtype, variant
        dev = "cpu"                                            # diffusers loads on the host; .to(device) moves
        self.unet = UNet2DConditionModel(unet_config_from_dict(_json(os.path.join(path, "unet", "config.json"))), dtype=dtype, device=dev)
        loaders.load_model_weights(self.unet, _weights_file(os.path.join(path, "unet"), "diffusion_pytorch_model", variant))
        vc = _json(os.path.join(path, "vae", "config.json"))
        self.vae_config = vc
        vcfg = VaeConfig(latent_channels=vc.get("latent_channels", 4), out_channels=vc.get("out_channels", 3),
                         block_out_channels=tuple(vc["block_out_channels"]), layers_per_block=vc.get("layers_per_block", 2),
                         norm_num_groups=vc.get("norm_num_groups", 32), scaling_factor=vc.get("scaling_factor", 0.13025))
        # `needs_upcasting = vae.dtype == float16 and vae.config.force_upcast` (lora_pipeline.py:641): the reference then decodes with
        # fp32 up blocks; AutoencoderKLDecoder(upcast=True) is that decode
        up = dtype == torch.float16 and bool(vc.get("force_upcast", True))
        self.vae = AutoencoderKLDecoder(vcfg, dtype=dtype, device=dev, upcast=up)
        loaders.load_model_weights(self.vae, _weights_file(os.path.join(path, "vae"), "diffusion_pytorch_model", variant), allow_extra=True)
        self.text_encoder = self._text_encoder("text_encoder", with_projection=False)
        self.text_encoder_2 = self._text_encoder("text_encoder_2", with_projection=True)
        from transformers import CLIPTokenizer
        self.tokenizer = CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer"))
        t2 = os.path.join(path, "tokenizer_2")
        self.tokenizer_2 = CLIPTokenizer.from_pretrained(t2) if os.path.isdir(t2) else self.tokenizer
        sched = os.path.join(path, "scheduler", "scheduler_config.json")
        name = _json(sched).get("_class_name", "EulerDiscreteScheduler") if os.path.exists(sched) else "EulerDiscreteScheduler"
        self.scheduler_class = DDIMScheduler if "DDIM" in name else EulerDiscreteScheduler     # SDXL-base ships EulerDiscrete
        self.bank = LoraBank(self.unet, [])

    def _text_encoder(self, sub: str, with_projection: bool) -> ClipTextEncoder:
        folder = os.path.join(self.path, sub)
        c = _json(os.path.join(folder, "config.json"))
        enc = ClipTextEncoder(_clip_config(c, with_projection or "WithProjection" in "".join(c.get("architectures", []))), dtype=self.dtype, device="cpu")
        loaders.load_model_weights(enc, _weights_file(folder, "model", self.variant), allow_extra=True)
        return enc

    def modules(self):
        return [self.unet, self.vae, self.text_encoder, self.text_encoder_2]


_CACHE: Dict[Tuple[str, str], _Components] = {}
_PRECISION_VARIANTS = {None, "fp16", "bf16", "fp32"}         # the same weights stored at another precision


def _components(path: str, dtype: torch.dtype, variant: Optional[str]) -> _Components:
    """One set of modules per (directory, dtype).  ``variant`` only chooses which file the FIRST load reads: diffusers' variants of a
    directory are the same weights stored at another precision, and the reference's scripts mix them (inference_instantid.py:187-201
    loads the main pipe with variant="fp16" and the concept pipe without) while the step engine needs ONE UNet behind both pipes."""
    key = (os.path.realpath(path), str(dtype))
    if key not in _CACHE:
        _CACHE[key] = _Components(path, dtype, variant)
    elif variant != _CACHE[key].variant and not {variant, _CACHE[key].variant} <= _PRECISION_VARIANTS:
        # "ema" and the like are OTHER weights, not the same weights at another precision: say so instead of silently sharing
        import warnings
        warnings.warn(f"{path}: already loaded with variant={_CACHE[key].variant!r}; the request for variant={variant!r} shares those modules "
                      "(omg_amd.compat keeps ONE set of modules per directory and dtype — call clear_component_cache() to load the other files)",
                      stacklevel=3)
    return _CACHE[key]


def clear_component_cache() -> None:
    _CACHE.clear()


def _tokenize(tok):
    def fn(prompts: Sequence[str]) -> torch.Tensor:
        return tok(list(prompts), padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
    return fn


def _cond_image_tensor(image, height: int, width: int) -> torch.Tensor:
    """``prepare_image``: PIL / array / tensor (or a list of them, one per sample of the request) -> (1, 3, H, W) in [0, 1]."""
    def one(im):
        if torch.is_tensor(im):
            t = im.float()
            return t if t.dim() == 4 else t[None]
        if hasattr(im, "resize"):                               # PIL
            im = im.convert("RGB").resize((width, height))
        a = np.asarray(im).astype(np.float32) / 255.0
        return torch.from_numpy(a).permute(2, 0, 1)[None]
    ims = [one(i) for i in image] if isinstance(image, (list, tuple)) else [one(image)]
    if any(not torch.equal(ims[0], t) for t in ims[1:]):
        raise L.OmgHipError("the samples of one request must share one spatial condition image (the reference passes [cond] * 2, "
                            "inference_lora.py:52-54)")
    return ims[0]


# ------------------------------------------------------------------------------------------------ models
class ControlNetModel(_ControlNetModel):
    @classmethod
    def from_pretrained(cls, path: str, torch_dtype: torch.dtype = torch.float16, variant: Optional[str] = None, **kw) -> "ControlNetModel":
        c = _json(os.path.join(path, "config.json"))
        net = cls(unet_config_from_dict({**c, "up_block_types": c.get("up_block_types", ())}), dtype=torch_dtype, device="cpu",
                  conditioning_channels=c.get("conditioning_channels", 3))
        loaders.load_model_weights(net, _weights_file(path, "diffusion_pytorch_model", variant))
        return net


class _PipeMixin:
    """from_pretrained / to / load_lora_weights / attribute surface shared by the main and the concept pipeline."""

    def _attach(self, comp: _Components) -> None:
        self._comp = comp
        self.vae, self.text_encoder, self.text_encoder_2 = comp.vae, comp.text_encoder, comp.text_encoder_2
        self.tokenizer, self.tokenizer_2 = comp.tokenizer, comp.tokenizer_2

    def to(self, device=None, dtype=None):
        if dtype is not None and dtype != self._comp.dtype:
            raise L.OmgHipError("load with torch_dtype=...; casting a built pipeline is not supported")
        if device is not None:
            for m in self._comp.modules():
                m.to(device)
            for extra in (getattr(self, "controlnet", None), getattr(self, "controlnet2", None), getattr(self, "image_proj_model", None)):
                if isinstance(extra, torch.nn.Module):
                    extra.to(device)
        return self

    @property
    def device(self):
        return self._comp.unet.device

    def enable_xformers_memory_efficient_attention(self, *a, **k) -> None:
        """No-op: every attention of this backend already is the fused flash kernel (the reference enables xformers on the concept
        pipe only, inference_lora.py:160)."""

    def load_lora_weights(self, path, weight_name: Optional[str] = None, adapter_name: Optional[str] = None, **kw) -> None:
        """``pipe.load_lora_weights(dir_or_file, weight_name="pytorch_lora_weights.safetensors", adapter_name=...)``
        (inference_lora.py:163-169): UNet half -> a named adapter of the shared LoraBank; text-encoder halves ride along."""
        p = os.fspath(path)
        if os.path.isdir(p):
            p = os.path.join(p, weight_name or "pytorch_lora_weights.safetensors")
        name = adapter_name or os.path.basename(os.fspath(path).rstrip("/")).split(".")[0]
        ad = loaders.load_lora_adapter(self._comp.unet, p, name)
        self._comp.bank.adapters[name] = ad
        self._comp.bank.version += 1
        self._comp.bank.slots = []                              # force a rebuild at the next call
        loaded = self.__dict__.setdefault("_loaded_adapters", [])
        if name not in loaded:
            loaded.append(name)

    def peft_active_adapters(self) -> List[Tuple[str, float]]:
        """What PEFT has switched on in THIS pipe when the loop never calls ``set_adapters``: ``load_lora_weights`` leaves the first
        adapter loaded into a pipe active at weight 1.0 (later ones are injected inactive); an explicit ``set_adapters`` replaces that.
        The InstantID flow relies on it — inference_instantid.py:220-222 loads a style LoRA into both pipes and
        instantid_pipeline.py never selects adapters — the LoRA flow does not (lora_pipeline.py:339-342, :588-591 select explicitly)."""
        explicit = getattr(self, "_active", ())
        if explicit:
            return [(n, float(w)) for n, w in explicit]
        loaded = self.__dict__.get("_loaded_adapters", [])
        return [(loaded[0], 1.0)] if loaded else []


class StableDiffusionXLPipeline(_PipeMixin, ConceptModels):
    """The reference's ``pipe_concept`` (inference_lora.py:159-170): UNet call, ``set_adapters``, ``encode_prompt``."""

    def __init__(self, comp: _Components):
        ConceptModels.__init__(self, comp.unet, comp.bank)
        self._attach(comp)
        self._encode = make_encode_prompt(comp.text_encoder, comp.text_encoder_2, _tokenize(comp.tokenizer), _tokenize(comp.tokenizer_2),
                                          adapters=comp.bank.adapters)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype: torch.dtype = torch.float16, variant: Optional[str] = None, controlnet=None, **kw):
        pipe = cls(_components(path, torch_dtype, variant))
        pipe.controlnet = controlnet      # inference_instantid.py:196-201 hands the concept pipe an IdentityNet it never uses; kept for `.to`
        return pipe

    def encode_prompt(self, prompt, prompt_2=None, device=None, num_images_per_prompt: int = 1, do_classifier_free_guidance: bool = True,
                      negative_prompt=None, negative_prompt_2=None, lora_scale: Optional[float] = None, **kw):
        """diffusers' return order: (prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds); the
        adapters chosen by ``set_adapters`` are active on both text encoders, scaled by ``lora_scale`` (lora_pipeline.py:336-347)."""
        if num_images_per_prompt != 1 or prompt_2 is not None or negative_prompt_2 is not None:
            raise L.OmgHipError("num_images_per_prompt != 1 / prompt_2 are not used by OMG and are not supported")
        return self._encode(prompt, negative_prompt if do_classifier_free_guidance else None, list(self._active), lora_scale)


class LoraMultiConceptPipeline(_PipeMixin, _LoraPipe):
    """``LoraMultiConceptPipeline.from_pretrained(...)`` (inference_lora.py:153-155) on :class:`omg_amd.pipeline.LoraMultiConceptPipeline`.
    ``__call__`` takes the reference's arguments (``prompt=[[p, p], [(rp, rn), ...]]``, ``negative_prompt=[n, n]``, ``image=``
    None | [pil, pil], ``output_type="pil"`` by default) and returns ``.images`` as two PIL images."""

    def __init__(self, comp: _Components, controlnet=None, scheduler=None):
        enc = make_encode_prompt(comp.text_encoder, comp.text_encoder_2, _tokenize(comp.tokenizer), _tokenize(comp.tokenizer_2),
                                 adapters=comp.bank.adapters)
        _LoraPipe.__init__(self, comp.unet, scheduler or comp.scheduler_class(), encode_prompt=enc, vae_decode=comp.vae.decode_latents)
        self._attach(comp)
        self.controlnet = controlnet

    @classmethod
    def from_pretrained(cls, path: str, controlnet=None, torch_dtype: torch.dtype = torch.float16, variant: Optional[str] = None, **kw):
        return cls(_components(path, torch_dtype, variant), controlnet=controlnet)

    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, image=None, height=None, width=None, output_type: str = "pil", return_dict: bool = True, **kw):
        if image is not None:
            h = height or self.unet.config.sample_size * self.vae_scale_factor
            w = width or self.unet.config.sample_size * self.vae_scale_factor
            cn = kw.get("controlnet", self.controlnet)
            n_nets = len(cn.nets) if hasattr(cn, "nets") else len(cn) if isinstance(cn, (list, tuple)) else 1
            if n_nets > 1:      # a list of ControlNets (the reference wraps it in MultiControlNetModel, lora_pipeline.py:175-176): one image (or [img] * 2) per net
                if not isinstance(image, (list, tuple)) or len(image) != n_nets:
                    raise L.OmgHipError(f"{n_nets} ControlNets need a list of {n_nets} conditioning images (lora_pipeline.py:366-383)")
                image = [_cond_image_tensor(im, h, w) for im in image]
            else:
                image = _cond_image_tensor(image, h, w)
        return _LoraPipe.__call__(self, prompt=prompt, prompt_2=prompt_2, image=image, height=height, width=width,
                                  output_type=output_type, return_dict=return_dict, **kw)


# ------------------------------------------------------------------------------------------------ InstantID
def load_image(path_or_image):
    """diffusers.utils.load_image: path / PIL -> RGB PIL (EXIF orientation applied)."""
    from PIL import Image, ImageOps
    im = Image.open(path_or_image) if isinstance(path_or_image, (str, os.PathLike)) else path_or_image
    return ImageOps.exif_transpose(im).convert("RGB")


def get_face_embedding(face_app, ref_images) -> list:
    """src/pipelines/instantid_pipeline.py:757-768: for each reference image the ArcFace embedding of ONE detected face — the first of
    the faces sorted ascending by the reference's key ``(x1 - x0) * y1 - y0`` (its comment says "maximum face"; the expression and
    the ascending sort are kept as they are).  ``face_app.get`` receives the image in BGR like ``cv2.cvtColor(..., COLOR_RGB2BGR)``."""
    embs = []
    for p in ref_images:
        rgb = np.array(load_image(p))
        info = face_app.get(np.ascontiguousarray(rgb[:, :, ::-1]))
        info = sorted(info, key=lambda x: (x["bbox"][2] - x["bbox"][0]) * x["bbox"][3] - x["bbox"][1])[0]
        embs.append(info["embedding"])
    return embs


class StableDiffusionXLInstantIDPipeline(StableDiffusionXLPipeline):
    """The container role the reference gives ``instantid_single_pieline.StableDiffusionXLInstantIDPipeline``:
    ``load_ip_adapter_instantid`` (:159-161) = ``set_image_proj_model`` (:163-184) + ``set_ip_adapter`` (:186-213),
    ``set_ip_adapter_scale`` (:215-219), ``_encode_prompt_image_emb`` (:221-243).  Its own ``__call__`` is dead in OMG's flows."""

    def load_ip_adapter_instantid(self, model_ckpt, image_emb_dim: int = 512, num_tokens: int = 16, scale: float = 0.5) -> None:
        self.set_image_proj_model(model_ckpt, image_emb_dim, num_tokens)
        self.set_ip_adapter(model_ckpt, num_tokens, scale)

    @staticmethod
    def _ckpt(model_ckpt) -> dict:
        return model_ckpt if isinstance(model_ckpt, dict) else torch.load(os.fspath(model_ckpt), map_location="cpu", weights_only=True)

    def set_image_proj_model(self, model_ckpt, image_emb_dim: int = 512, num_tokens: int = 16) -> None:
        unet = self._comp.unet
        self.image_proj_model = Resampler(dim=1280, depth=4, dim_head=64, heads=20, num_queries=num_tokens, embedding_dim=image_emb_dim,
                                          output_dim=unet.config.cross_attention_dim, ff_mult=4, dtype=unet.dtype, device=unet.device)
        sd = self._ckpt(model_ckpt)
        if "image_proj" in sd:
            sd = sd["image_proj"]
        self.image_proj_model.load_state_dict({k: v.to(unet.dtype) for k, v in sd.items()})
        self.image_proj_model_in_features = image_emb_dim

    def set_ip_adapter(self, model_ckpt, num_tokens: int = 16, scale: float = 0.5) -> None:
        self.ip_adapter = IPAdapter(self._comp.unet, num_tokens=num_tokens, scale=scale)
        self.ip_adapter.load_state_dict(self._ckpt(model_ckpt))

    def set_ip_adapter_scale(self, scale: float) -> None:
        self.ip_adapter.set_scale(scale)

    @torch.no_grad()
    def _encode_prompt_image_emb(self, prompt_image_emb, device=None, num_images_per_prompt: int = 1, dtype=None,
                                 do_classifier_free_guidance: bool = True) -> torch.Tensor:
        unet = self._comp.unet
        e = torch.as_tensor(np.asarray(prompt_image_emb) if not torch.is_tensor(prompt_image_emb) else prompt_image_emb)
        e = e.to(device=unet.device, dtype=unet.dtype).reshape(1, -1, self.image_proj_model_in_features)
        if do_classifier_free_guidance:
            e = torch.cat([torch.zeros_like(e), e], dim=0)
        return self.image_proj_model(e)                         # (2, 16, Cx): [tokens of the zero embedding, tokens of the identity]


class InstantidMultiConceptPipeline(_PipeMixin, _InstantidPipe):
    """``InstantidMultiConceptPipeline.from_pretrained(model, controlnet=identitynet, ...)`` and the reference's ``__call__``
    arguments (instantid_pipeline.py:212-258): ``prompt=[[p, p], [(rp, rn, ref_image_path), ...]]``, ``image`` = key-point image
    (stage 2 only), ``t2i_image`` / ``t2i_controlnet_conditioning_scale`` for ``pipe.controlnet2``, ``face_app``; the face
    embeddings go through ``concept_models._encode_prompt_image_emb`` inside the call (:378-388)."""

    def __init__(self, comp: _Components, controlnet=None, scheduler=None):
        enc = make_encode_prompt(comp.text_encoder, comp.text_encoder_2, _tokenize(comp.tokenizer), _tokenize(comp.tokenizer_2),
                                 adapters=comp.bank.adapters)
        _InstantidPipe.__init__(self, comp.unet, controlnet, scheduler or comp.scheduler_class(), encode_prompt=enc,
                                vae_decode=comp.vae.decode_latents)
        self._attach(comp)

    @classmethod
    def from_pretrained(cls, path: str, controlnet=None, torch_dtype: torch.dtype = torch.float16, variant: Optional[str] = None, **kw):
        return cls(_components(path, torch_dtype, variant), controlnet=controlnet)

    @torch.no_grad()
    def __call__(self, prompt=None, negative_prompt=None, image=None, t2i_image=None, height=None, width=None, face_app=None,
                 concept_models=None, stage=None, cross_attention_kwargs=None, output_type: str = "pil", return_dict: bool = True,
                 indices_to_alter=None, **kw):
        h = height or self.unet.config.sample_size * self.vae_scale_factor
        w = width or self.unet.config.sample_size * self.vae_scale_factor
        extra = {}
        if prompt is not None:
            # all prompts are encoded at once by the MAIN pipe (instantid_pipeline.py:336-375), no LoRA involved
            # ... but a LoRA that load_lora_weights left active on the main pipe (inference_instantid.py:220-222) acts on its text
            # encoders too, at lora_scale = cross_attention_kwargs["scale"] (instantid_pipeline.py:330-360)
            te_scale = (cross_attention_kwargs or {}).get("scale", None)
            glob, regions = list(prompt[0]), list(prompt[1])
            # prompt_2 / negative_prompt_2 / clip_skip go to the ONE encode_prompt call over all prompts (instantid_pipeline.py:343-362)
            enc_extra = {k: kw.pop(k) for k in ("prompt_2", "negative_prompt_2", "clip_skip") if kw.get(k) is not None}
            pe, ne, pp, npp = self.encode_prompt(glob + [r[0] for r in regions], list(negative_prompt) + [r[1] for r in regions],
                                                 self.peft_active_adapters() or None, te_scale, **enc_extra)
            extra = dict(prompt_embeds=pe[:2], negative_prompt_embeds=ne[:2], pooled_prompt_embeds=pp[:2], negative_pooled_prompt_embeds=npp[:2],
                         region_prompt_embeds=[(ne[2 + c: 3 + c], pe[2 + c: 3 + c], npp[2 + c: 3 + c], pp[2 + c: 3 + c]) for c in range(len(regions))])
            if stage == 2:
                embs = get_face_embedding(face_app, [r[2] for r in regions])
                extra["region_image_embeds"] = [concept_models._encode_prompt_image_emb(e, concept_models._execution_device, 1,
                                                                                        concept_models._unet.dtype, True) for e in embs]
        if image is not None:
            image = _cond_image_tensor(image, h, w)
        if t2i_image is not None:
            t2i_image = _cond_image_tensor(t2i_image, h, w)
        # PEFT's "whatever load_lora_weights left switched on": main rows at cross_attention_kwargs["scale"] (:596-616), concept rows at 1.0
        # (the concept UNet is called with cross_attention_kwargs=None, :665-674)
        c_act = concept_models.peft_active_adapters() if hasattr(concept_models, "peft_active_adapters") else []
        return _InstantidPipe.__call__(self, image=image, t2i_image=t2i_image, height=height, width=width, concept_models=concept_models,
                                       stage=stage, output_type=output_type, return_dict=return_dict, cross_attention_kwargs=cross_attention_kwargs,
                                       main_adapters=self.peft_active_adapters() or None, concept_adapters=c_act or None, **extra, **kw)


# ------------------------------------------------------------------------------------------------ alias modules
InstantidSingleConceptPipeline = StableDiffusionXLInstantIDPipeline      # the name inference_instantid.py:35 imports


class _Unavailable:
    """Stand-in for a class the scripts import but this backend does not provide (imported-but-unused names such as
    ``DPMSolverMultistepScheduler`` at inference_instantid.py:8, or third-party models outside the hot path): importing is fine,
    using it says what is missing."""

    def __init__(self, name: str):
        self._name = name

    def __call__(self, *a, **k):
        raise L.OmgHipError(f"{self._name} is not provided by omg_amd.compat (it is outside the denoising hot path); install the real package")

    def __getattr__(self, attr):
        raise L.OmgHipError(f"{self._name}.{attr}: {self._name} is not provided by omg_amd.compat; install the real package")


def _save_image(tensor, fp, nrow: int = 8, **kw) -> None:
    """``torchvision.utils.save_image`` for the one way the scripts use it (a (3, H, W) or (1, 3, H, W) tensor in [0, 1] to a file)."""
    from PIL import Image
    t = torch.as_tensor(tensor).detach().float().cpu()
    if t.dim() == 4:
        t = torch.cat(list(t), dim=2)                            # a row of images
    if t.dim() == 2:
        t = t[None]
    a = (t.clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).numpy()
    Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a).save(fp)


_SAVED_DIFFUSERS: dict = {}      # a real `diffusers` set aside by install(), put back by uninstall()


def install(stub_missing: bool = True) -> List[str]:
    """Make the reference's scripts run on this backend with their ORIGINAL import block (B3): registers, in ``sys.modules``, the module
    names ``inference_lora.py:29-32`` and ``inference_instantid.py:8-13, :34-37`` import, backed by this package —

        src.pipelines.lora_pipeline          LoraMultiConceptPipeline, revise_regionally_controlnet_forward
        src.pipelines.instantid_pipeline     InstantidMultiConceptPipeline, revise_regionally_controlnet_forward
        src.pipelines.instantid_single_pieline   InstantidSingleConceptPipeline
        src.prompt_attention.p2p_attention   AttentionReplace
        diffusers                            ControlNetModel, StableDiffusionXLPipeline (+ DPMSolverMultistepScheduler, models.T2IAdapter:
                                             imported by the InstantID script and never used — placeholders), utils.load_image

    so ``import omg_amd.compat as c; c.install()`` in front of the script (or ``python -c "import omg_amd.compat as c; c.install();
    import runpy; runpy.run_path('inference_lora.py', run_name='__main__')"``) is the whole edit.  Call it BEFORE the script's imports; it
    overrides a real ``diffusers`` for the process (that is its purpose).  ``stub_missing``: third-party modules of the import block that
    are outside the hot path and absent from the environment get stand-ins — ``torchvision.utils.save_image`` (a real implementation
    on PIL), ``cv2`` / ``insightface.app.FaceAnalysis`` (importable; using them raises).  Returns the names it registered."""
    import importlib
    import importlib.util
    import sys
    import types
    from .controller import AttentionReplace
    from .pipeline import revise_regionally_controlnet_forward

    done: List[str] = []

    def package(name: str):
        """a real (namespace) package of that name stays; otherwise an empty stand-in package"""
        if name in sys.modules:
            return sys.modules[name]
        try:
            if importlib.util.find_spec(name) is not None:
                return importlib.import_module(name)
        except (ImportError, ValueError):
            pass
        m = types.ModuleType(name)
        # a package: submodules are looked up in sys.modules first (the aliases), then in the directories of that name found on sys.path NOW —
        # a checkout's own `src/efficientvit`, `src/...` stay importable beside the aliased `src.pipelines.*` (an empty __path__
        # shadowed them whenever the stand-in was created before the script's directory was on sys.path)
        m.__path__ = [d for d in (os.path.join(p or os.getcwd(), *name.split(".")) for p in sys.path) if os.path.isdir(d)]
        m.__omg_amd_alias__ = True
        sys.modules[name] = m
        done.append(name)
        return m

    def module(name: str, **attrs):
        parent = name.rpartition(".")[0]
        if parent:
            package(parent.partition(".")[0])
            acc = ""
            for part in parent.split("."):
                acc = part if not acc else acc + "." + part
                package(acc)
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__omg_amd_alias__ = True
        sys.modules[name] = m
        if parent and isinstance(sys.modules.get(parent), types.ModuleType):
            try:
                setattr(sys.modules[parent], name.rpartition(".")[2], m)
            except Exception:
                pass
        done.append(name)
        return m

    module("src.pipelines.lora_pipeline", LoraMultiConceptPipeline=LoraMultiConceptPipeline,
           revise_regionally_controlnet_forward=revise_regionally_controlnet_forward)
    module("src.pipelines.instantid_pipeline", InstantidMultiConceptPipeline=InstantidMultiConceptPipeline,
           revise_regionally_controlnet_forward=revise_regionally_controlnet_forward)
    module("src.pipelines.instantid_single_pieline", InstantidSingleConceptPipeline=InstantidSingleConceptPipeline,
           StableDiffusionXLInstantIDPipeline=StableDiffusionXLInstantIDPipeline)
    module("src.prompt_attention.p2p_attention", AttentionReplace=AttentionReplace)
    # a REAL diffusers (none in this image; a user's environment may have one) is set aside whole and restored by uninstall(): the aliases never
    # patch it (package("diffusers") imported the real package and overwrote four of its attributes for good)
    # MERGED, not overwritten: a second install() without an uninstall() in between finds only the aliases, and assigning that empty result dropped the
    # real package the first call had set aside for good
    _SAVED_DIFFUSERS.update({n: sys.modules.pop(n) for n in [n for n in sys.modules if n == "diffusers" or n.startswith("diffusers.")]
                             if not getattr(sys.modules[n], "__dict__", {}).get("__omg_amd_alias__", False)})
    d = types.ModuleType("diffusers")
    d.__path__ = []
    d.__omg_amd_alias__ = True
    sys.modules["diffusers"] = d
    done.append("diffusers")
    dm = module("diffusers.models", ControlNetModel=ControlNetModel, T2IAdapter=_Unavailable("diffusers.models.T2IAdapter"))
    du = module("diffusers.utils", load_image=load_image)
    d.__dict__.update(ControlNetModel=ControlNetModel, StableDiffusionXLPipeline=StableDiffusionXLPipeline,
                      DPMSolverMultistepScheduler=_Unavailable("diffusers.DPMSolverMultistepScheduler"),
                      DDIMScheduler=DDIMScheduler, EulerDiscreteScheduler=EulerDiscreteScheduler, models=dm, utils=du)
    if stub_missing:
        def absent(name: str) -> bool:
            if name in sys.modules:
                return False
            try:
                return importlib.util.find_spec(name) is None
            except (ImportError, ValueError):
                return True
        if absent("torchvision"):
            module("torchvision.utils", save_image=_save_image)
        if absent("cv2"):
            cv2 = module("cv2")
            cv2.__getattr__ = lambda a: _Unavailable("cv2").__getattr__(a)          # PEP 562: any use says what is missing
        if absent("insightface"):
            module("insightface.app", FaceAnalysis=_Unavailable("insightface.app.FaceAnalysis"))
    return done


def uninstall() -> None:
    """Remove what :func:`install` registered (names already imported from the aliases stay bound where they were imported)."""
    import sys
    # vars(), not getattr: lazy modules (transformers) import on attribute access and would grow sys.modules under the iteration
    for n in [n for n, m in list(sys.modules.items()) if m is not None and getattr(m, "__dict__", {}).get("__omg_amd_alias__", False)]:
        del sys.modules[n]
    global _SAVED_DIFFUSERS
    sys.modules.update(_SAVED_DIFFUSERS)
    _SAVED_DIFFUSERS = {}
