"""OMG's two-stage denoising loop on the MI355X — boundary B3 of SURVEY.md §8b.

Mirrors ``LoraMultiConceptPipeline.__call__`` (/root/reference src/pipelines/lora_pipeline.py:212-669):
same keyword names and meaning for everything on the hot path (``num_inference_steps``,
``guidance_scale``, ``generator``/``latents``, ``cross_attention_kwargs={"scale": 0.8}``, ``controller``,
``concept_models``, ``stage`` in {1, 2}, ``region_masks``, ``lora_list``, ``styleL``), and the same batch
layout: latents duplicated x2 (:409), model input ``[unc0, unc1, cond0, cond1]`` (:467-474, :491), concept
pass on ``latent_model_input[3:4]`` duplicated (:583-585), fusion for ``i > 15 and stage == 2`` (:568).

Text encoders and the VAE are separate modules (rows N4 / N1 of SURVEY §8f).  Without them the call takes ``prompt_embeds`` /
pooled embeddings (the reference computes them at :315-347 and passes them on) and returns latents
(``output_type="latent"``), or decoded images (``"pil"``, the reference's default, / ``"pt"`` / ``"np"``) when the pipeline was built with
``vae_decode=omg_amd.vae.AutoencoderKLDecoder(...).decode_latents`` (row N1; the reference's tail at :635-661);
``prompt=`` works when the pipeline was built with ``encode_prompt=omg_amd.text_encoder.make_encode_prompt(...)``.

MI355X-first differences from the reference's loop, all value-preserving:
  * the K per-concept UNet passes of a step run as ONE batched forward with a per-sample LoRA slot;
  * time/text-conditioning embeddings of all steps are computed once before the loop;
  * fusion + CFG + scheduler step + next model input are one kernel, masks stay on the device
    (the reference copies each mask to the host every step and boolean-indexes, :572-602, :675-679);
  * each step regime (plain / fused, self-replace on / off) can be captured as a hipGraph.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import ops
from .attention import Attention, RegionControlNet_AttnProcessor
from .lora import LoraBank
from .modules import LoraState, pointer_epoch
from .schedulers import DDIMScheduler

FUSION_START = 15   # `if i > 15 and stage == 2` (lora_pipeline.py:568) — absolute, not relative
CONCEPT_LORA_SCALE = 0.8   # `cross_attention_kwargs={'scale': 0.8}` of every concept UNet call (lora_pipeline.py:596), whatever the caller passes


def controlnet_keep(n_steps: int, start, end, n_nets: int = 1) -> List[List[float]]:
    """``controlnet_keep`` of the reference (lora_pipeline.py:275-286 broadcast + :421-428): for step i and net k the factor on the conditioning scale is
    1.0 while  start_k <= i / S  and  (i + 1) / S <= end_k,  else 0.0.  ``start`` / ``end``: scalars (every net) or one entry per net."""
    def per_net(v, other):
        if isinstance(v, (list, tuple)):
            return [float(x) for x in v]
        return [float(v)] * (len(other) if isinstance(other, (list, tuple)) else n_nets)
    starts, ends = per_net(start, end), per_net(end, start)
    if len(starts) != len(ends):
        raise ValueError(f"control_guidance_start has {len(starts)} entries, control_guidance_end {len(ends)}")
    return [[0.0 if (i / n_steps < s_ or (i + 1) / n_steps > e_) else 1.0 for s_, e_ in zip(starts, ends)] for i in range(n_steps)]


def revise_regionally_controlnet_forward(unet, controller) -> None:
    """Install the region/controller attention processor on every ``Attention`` of the UNet and set
    ``controller.num_att_layers`` — same name, traversal order and place labels (including the
    mid/up label swap, harmless because the label is unused) as lora_pipeline.py:136-152."""

    def change_forward(module, count, place_in_unet):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == "Attention":
                layer.set_processor(RegionControlNet_AttnProcessor(controller=controller, place_in_unet=place_in_unet))
                if "attn2" in name:
                    count += 1
            else:
                count = change_forward(layer, count, place_in_unet)
        return count

    n = change_forward(unet.down_blocks, 0, "down")
    n = change_forward(unet.mid_block, n, "up")
    n = change_forward(unet.up_blocks, n, "mid")
    print(f"Number of attention layer registered {n}")
    controller.num_att_layers = n * 2


class ConceptModels:
    """What the reference's loop needs from ``concept_models`` (a second SDXL pipeline with the concept
    LoRAs loaded; inference_lora.py:159-170): ``unet(...)``, ``set_adapters(...)``, ``_execution_device``.
    Here the concept UNet shares the base weights of the main UNet; adapters live in a :class:`LoraBank`."""

    def __init__(self, unet, bank: Optional[LoraBank] = None):
        self._unet = unet
        self.bank = bank
        self._active: Tuple[Tuple[str, float], ...] = ()
        self._state_cache: Dict[Tuple, LoraState] = {}

    @property
    def _execution_device(self):
        return self._unet.device

    def set_adapters(self, adapter_names, adapter_weights=None) -> None:
        if isinstance(adapter_names, str):
            adapter_names = [adapter_names]
        if adapter_weights is None:
            adapter_weights = [1.0] * len(adapter_names)
        self._active = tuple((n, float(w)) for n, w in zip(adapter_names, adapter_weights))

    def lora_state(self, slots: Sequence[int], merged: bool = False) -> Optional[LoraState]:
        """``slots[b]`` = LoRA slot of sample b.  segment mode: -1 = no adapter; merged mode: 0 = base weights,
        s+1 = bank slot s."""
        if self.bank is None:
            return None
        key = (tuple(slots), merged)
        st = self._state_cache.get(key)
        if st is None:
            st = LoraState(torch.tensor(list(slots), dtype=torch.int32, device=self._unet.device), len(slots), merged)
            self._state_cache[key] = st
        return st

    def unet_batched(self, sample, timestep, encoder_hidden_states, slots: Sequence[int], **kw):
        """Concept forward where sample b uses LoRA slot ``slots[b]``; bypasses the p2p controller."""
        cak = dict(kw.pop("cross_attention_kwargs", None) or {})
        cak["omg_bypass_controller"] = True
        merged = self.bank is not None and self.bank.mode == "merged"
        if merged:
            slots = [s + 1 for s in slots]            # slot 0 of the merged stacks is the base weight
        self._unet.set_lora_state(self.lora_state(slots, merged))
        try:
            return self._unet(sample, timestep, encoder_hidden_states=encoder_hidden_states, cross_attention_kwargs=cak, **kw)
        finally:
            self._unet.set_lora_state(None)

    def unet(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None, added_cond_kwargs=None,
             return_dict=False, **kw):
        """Reference call shape (lora_pipeline.py:592-599): whole batch uses the adapters chosen by ``set_adapters``."""
        slot = self.bank.slot_of(self._active) if self.bank is not None and self._active else -1
        return self.unet_batched(sample, timestep, encoder_hidden_states, [slot] * sample.shape[0],
                                 cross_attention_kwargs=cross_attention_kwargs, added_cond_kwargs=added_cond_kwargs,
                                 return_dict=return_dict, **kw)


class StageCache:
    """Exact redundancy BETWEEN the two calls of one image (OFF unless a cache object is passed; "flag when used").

    The reference's flow is stage 1 -> segment -> stage 2 with the SAME seed, prompts and kwargs (inference_lora.py:262-297), and the
    fusion branch only fires for ``i > 15`` (lora_pipeline.py:568): steps 0..15 of the stage-2 call repeat steps 0..15 of the stage-1
    call operation for operation.  A call with ``stage_cache=cache`` that runs those steps stores, per request, the latents ENTERING
    the first fused step under a key made of everything that determines them (initial latents, all main-pass embeddings, step count,
    guidance, scheduler table, size, adapters, ControlNet + image, UNet identity and weight version); a later stage-2 call that finds
    every request's key starts at step 16 from the stored latents — 16 x 4 sample-forwards per image fewer, bitwise the same result
    with this package's batch-invariant deterministic kernels (tests/test_pipeline_gpu.py).  Nothing is assumed: a miss runs the full call."""

    def __init__(self, max_entries: int = 64):
        self.entries: Dict[str, torch.Tensor] = {}
        self.base: Dict[str, Dict[int, Tuple[torch.Tensor, torch.Tensor]]] = {}
        self.max_entries = max_entries
        self.hits = 0
        self.misses = 0

    @staticmethod
    def digest(*parts) -> str:
        import hashlib
        h = hashlib.sha1()
        for p in parts:
            if torch.is_tensor(p):
                h.update(str((tuple(p.shape), str(p.dtype))).encode())
                h.update(p.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes())
            else:
                h.update(repr(p).encode())
        return h.hexdigest()

    def put(self, key: str, latents: torch.Tensor) -> None:
        if key not in self.entries and len(self.entries) >= self.max_entries:
            old = next(iter(self.entries))
            self.entries.pop(old)
            self.base.pop(old, None)
        self.entries[key] = latents.detach().clone()

    def get(self, key: str) -> Optional[torch.Tensor]:
        return self.entries.get(key)

    # ---- the BASE sample's trajectory behind the first fused step (last item): sample 0 of a call is never fused
    # (lora_pipeline.py:605-607 write samples 1 and 3 only), so its latents at every step are those of the stage-1 call.  With them
    # cached the stage-2 call need not compute `unc0` at all (its prediction only feeds sample 0's update); `cond0` stays — `cond1`
    # borrows its attention probabilities — and is fed the cached latents.
    def put_base(self, key: str, k: int, latents0: torch.Tensor, model_input0: torch.Tensor) -> None:
        """sample 0 ENTERING step k: its latents (fp32) and the scaled model input the step kernel wrote for it"""
        self.base.setdefault(key, {})[k] = (latents0.detach().clone(), model_input0.detach().clone())

    def get_base(self, key: str, first: int, n_steps: int):
        """{k: (latents0, model_input0)} for k = first + 1 .. n_steps if every step is there, else None"""
        b = self.base.get(key)
        if b is None or any(k not in b for k in range(first + 1, n_steps + 1)):
            return None
        return b


# Parameters of the reference's ``__call__`` (lora_pipeline.py:212-255, instantid_pipeline.py:215-259) that change ITS output when they leave their
# default and that this engine does not implement: name -> (the default, where the reference uses it).  A non-default value is REFUSED — ignoring
# it would return an image the reference would not have produced, without a word.
_UNIMPLEMENTED = {
    "num_images_per_prompt": (1, "lora_pipeline.py:223"), "ip_adapter_image": (None, "lora_pipeline.py:231"),
}
# Implemented in round 6 and therefore no longer here: guess_mode, control_guidance_start / _end, a list of ControlNets (generate_many), prompt_2 /
# negative_prompt_2 / clip_skip (omg_amd.text_encoder.make_encode_prompt), negative_original_size / _crops_coords_top_left / _target_size, callback /
# callback_steps / callback_on_step_end (+ _tensor_inputs).
# names the reference's own ``**kwargs`` swallows without reading (the shipped scripts pass them: inference_lora.py:241-245), or reads only
# together with a callback
_REFERENCE_IGNORES = ("spatial_condition", "indices_to_alter")
_CALLBACK_TENSORS = ("latents", "prompt_embeds", "negative_prompt_embeds")      # _callback_tensor_inputs (lora_pipeline.py:165)


def refuse_unimplemented(given: dict, leftovers: dict, who: str) -> None:
    """``given``: the reference-signature parameters this engine does not implement, as the caller passed them; ``leftovers``: what is
    still in ``**kwargs`` after the engine took its own.  Raises OmgHipError for a non-default reference parameter and for a name nobody knows."""
    for name, value in given.items():
        default, where = _UNIMPLEMENTED[name]
        v = value[0] if isinstance(value, (list, tuple)) and len(value) == 1 and not isinstance(default, tuple) else value
        v = tuple(v) if isinstance(default, tuple) and isinstance(v, (list, tuple)) else v
        if not (v is default or v == default):
            raise L.OmgHipError(f"{who}: {name}={value!r} is not implemented (the reference uses it at {where}); only the default {default!r} is")
    unknown = [k for k in leftovers if k not in _REFERENCE_IGNORES]
    if unknown:
        raise L.OmgHipError(f"{who}: unknown keyword argument(s) {sorted(unknown)} — neither a parameter of the reference's __call__ nor one of this engine's")


class StableDiffusionXLPipelineOutput(SimpleNamespace):
    pass


class LoraMultiConceptPipeline:
    """See the module docstring.  ``__call__`` keeps the reference's single-request signature; ``generate_many`` runs
    several independent requests in lock-step through ONE batched UNet forward per step (images are independent,
    SURVEY §8e) — same per-image arithmetic, larger GEMMs."""

    def __init__(self, unet, scheduler=None, encode_prompt: Optional[Callable] = None, vae_decode: Optional[Callable] = None):
        self.unet = unet
        self.scheduler = scheduler or DDIMScheduler()
        self.encode_prompt = encode_prompt
        self.vae_decode = vae_decode
        self.vae_scale_factor = 8
        self._engines: Dict[Tuple, SimpleNamespace] = {}      # call shape -> static buffers + captured step graphs, LRU
        self.max_engines = 8                                   # a few hundred MB of buffers each (+ graph pools): raise for servers with many shapes
        self.engine_evictions = 0

    @property
    def _execution_device(self):
        return self.unet.device

    # ------------------------------------------------------------------ helpers
    def prepare_latents(self, batch, channels, height, width, dtype, device, generator, latents=None):
        """randn / vae_scale_factor, times init_noise_sigma (diffusers prepare_latents; lora_pipeline.py:397-406).
        ``latents=`` injects fixed noise (needed for CPU<->GPU seed parity)."""
        shape = (batch, channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gen_dev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gen_dev, dtype=torch.float32).to(device)
        else:
            latents = latents.to(device=device, dtype=torch.float32)
            if tuple(latents.shape) != shape:
                raise ValueError(f"latents shape {tuple(latents.shape)} != {shape}")
        return latents * float(self.scheduler.init_noise_sigma)

    @staticmethod
    def _add_time_ids(original_size, crops, target_size, n, device):
        ids = list(original_size) + list(crops) + list(target_size)
        return torch.tensor([ids] * n, dtype=torch.float32, device=device)

    def _all_step_embeddings(self, ts: torch.Tensor, text: torch.Tensor, tids: torch.Tensor, unet=None) -> torch.Tensor:
        """emb[i] for every step in one batched pass: (S, B, 4*C0).  Depends only on (t_i, pooled text, time ids) and on the
        embedding layers of the UNet that will consume it (``unet``: the concept pipe's, when it is not the main one)."""
        S, B = ts.numel(), text.shape[0]
        t_all = ts.reshape(S, 1).expand(S, B).reshape(-1).contiguous()
        emb = (unet or self.unet).time_embed(t_all, S * B, text.repeat(S, 1), tids.repeat(S, 1))
        return emb.view(S, B, -1)

    # ------------------------------------------------------------------ the reference's call signature
    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, image=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 5.0, negative_prompt=None, negative_prompt_2=None,
                 num_images_per_prompt: int = 1, eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                 output_type: str = "pil", return_dict: bool = True, cross_attention_kwargs=None,
                 original_size=None, crops_coords_top_left=(0, 0), target_size=None,
                 controller=None, concept_models: Optional[ConceptModels] = None, stage: Optional[int] = None,
                 region_masks: Optional[Sequence[Optional[torch.Tensor]]] = None, lora_list: Optional[Sequence[str]] = None,
                 styleL: Optional[bool] = None, region_prompt_embeds: Optional[Sequence[Tuple[torch.Tensor, ...]]] = None,
                 use_graph: bool = False, trajectory: Optional[list] = None, fusion_start: int = FUSION_START,
                 lora_mode: str = "merged", dedup: bool = False, **kwargs):
        controlnet = kwargs.pop("controlnet", getattr(self, "controlnet", None))
        cn_scale = kwargs.pop("controlnet_conditioning_scale", 1.0)
        # round 6: implemented instead of refused (lora_pipeline.py:236-238; generate_many's docstring)
        guess_mode = bool(kwargs.pop("guess_mode", False))
        cg_start, cg_end = kwargs.pop("control_guidance_start", 0.0), kwargs.pop("control_guidance_end", 1.0)
        clip_skip = kwargs.pop("clip_skip", None)                                                   # lora_pipeline.py:245, :333
        neg_cond = (kwargs.pop("negative_original_size", None), kwargs.pop("negative_crops_coords_top_left", (0, 0)),
                    kwargs.pop("negative_target_size", None))                                       # :242-244, :459-466
        callbacks = dict(callback=kwargs.pop("callback", None), callback_steps=kwargs.pop("callback_steps", None),      # :256-257, :629-632
                         callback_on_step_end=kwargs.pop("callback_on_step_end", None),                                  # :246, :617-626
                         callback_on_step_end_tensor_inputs=kwargs.pop("callback_on_step_end_tensor_inputs", ("latents",)))
        stage_cache, drop_unc0 = kwargs.pop("stage_cache", None), kwargs.pop("drop_unc0", False)
        given = {k: kwargs.pop(k) for k in list(kwargs) if k in _UNIMPLEMENTED}
        given.update(num_images_per_prompt=num_images_per_prompt)
        refuse_unimplemented(given, kwargs, "LoraMultiConceptPipeline.__call__")
        n_nets = len(controlnet.nets) if hasattr(controlnet, "nets") else len(controlnet) if isinstance(controlnet, (list, tuple)) else 1
        if n_nets == 1:
            if isinstance(cn_scale, (list, tuple)):      # lora_pipeline.py:513-515: one ControlNet takes the first entry
                cn_scale = cn_scale[0]
            if isinstance(image, (list, tuple)):         # a one-element list reached generate_many un-unwrapped
                if len(image) != 1:
                    raise L.OmgHipError(f"{len(image)} ControlNet images for one ControlNet: pass one image, or a list of ControlNets (lora_pipeline.py:175-176)")
                image = image[0]
        elif image is not None and (not isinstance(image, (list, tuple)) or len(image) != n_nets):
            raise L.OmgHipError(f"{n_nets} ControlNets need a list of {n_nets} conditioning images (lora_pipeline.py:366-383)")
        if image is not None and controlnet is None:
            raise L.OmgHipError("image= needs a ControlNet: pass controlnet=omg_amd.controlnet.ControlNetModel(...)")
        if eta != 0.0:
            raise L.OmgHipError("eta != 0 (stochastic DDIM) is not used by OMG and is not supported")
        lora_list = list(lora_list or [])
        # ---- 3. prompt embeddings (global on the main pipe; per-region on the concept pipe)
        if prompt_embeds is None:
            if self.encode_prompt is None:
                raise L.OmgHipError("no text encoders attached: pass prompt_embeds=/pooled_prompt_embeds= (and region_prompt_embeds=) or "
                                    "construct the pipeline with encode_prompt=omg_amd.text_encoder.make_encode_prompt(...)")
            # global prompt on the main pipe (its text encoders carry the style LoRA when one is loaded, inference_lora.py:162-164),
            # region prompts on the concept pipe with the concept's adapters active (lora_pipeline.py:315-347); both with
            # lora_scale = cross_attention_kwargs["scale"]
            te_scale = (cross_attention_kwargs or {}).get("scale", None)
            global_prompt, regions = prompt[0], prompt[1]
            # prompt_2 / negative_prompt_2 / clip_skip act on the GLOBAL prompt only: the region prompts are encoded without them (lora_pipeline.py:340-342)
            extra = {k: v for k, v in (("prompt_2", prompt_2), ("negative_prompt_2", negative_prompt_2), ("clip_skip", clip_skip)) if v is not None}
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = self.encode_prompt(
                global_prompt, negative_prompt, [("style", 1.0)] if styleL else None, te_scale, **extra)
            region_prompt_embeds = []
            for lora_param, (rp, rn) in zip(lora_list, [(r[0], r[1]) for r in regions]):
                if hasattr(concept_models, "encode_prompt"):      # the reference's literal sequence (lora_pipeline.py:337-343)
                    if styleL:
                        concept_models.set_adapters([lora_param, "style"], adapter_weights=[0.7, 0.5])
                    else:
                        concept_models.set_adapters(lora_param)
                    pe, ne, pp, npp = concept_models.encode_prompt(prompt=rp, device=concept_models._execution_device, num_images_per_prompt=1,
                                                                   do_classifier_free_guidance=True, negative_prompt=rn, lora_scale=te_scale)
                else:
                    combo = [(lora_param, 0.7), ("style", 0.5)] if styleL else [(lora_param, 1.0)]
                    pe, ne, pp, npp = self.encode_prompt(rp, rn, combo, te_scale)
                region_prompt_embeds.append((ne, pe, npp, pp))
        req = dict(prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                   pooled_prompt_embeds=pooled_prompt_embeds, negative_pooled_prompt_embeds=negative_pooled_prompt_embeds,
                   region_prompt_embeds=region_prompt_embeds, region_masks=region_masks, latents=latents, generator=generator)
        traj_many = [] if trajectory is not None else None
        lat = self.generate_many([req], height=height, width=width, num_inference_steps=num_inference_steps,
                                 guidance_scale=guidance_scale, cross_attention_kwargs=cross_attention_kwargs,
                                 original_size=original_size, crops_coords_top_left=crops_coords_top_left, target_size=target_size,
                                 controller=controller, concept_models=concept_models, stage=stage, lora_list=lora_list,
                                 styleL=styleL, use_graph=use_graph, trajectory=traj_many, fusion_start=fusion_start,
                                 lora_mode=lora_mode, controlnet=controlnet if image is not None else None, controlnet_image=image,
                                 controlnet_conditioning_scale=cn_scale, dedup=dedup, stage_cache=stage_cache, drop_unc0=drop_unc0,
                                 guess_mode=guess_mode and image is not None, control_guidance_start=cg_start, control_guidance_end=cg_end,
                                 negative_original_size=neg_cond[0], negative_crops_coords_top_left=neg_cond[1], negative_target_size=neg_cond[2],
                                 **callbacks)[0]
        if trajectory is not None:
            trajectory.extend(t[0] for t in traj_many)
        images = self._postprocess(lat, output_type)
        if not return_dict:
            return (images,)
        return StableDiffusionXLPipelineOutput(images=images)

    def _postprocess(self, lat: torch.Tensor, output_type: str):
        """The reference's tail (lora_pipeline.py:635-661): VAE decode + ``image_processor.postprocess``.  ``"latent"``: the final
        latents; ``"pt"``: decoded (n, 3, H, W) in [0, 1]; ``"np"``; ``"pil"`` (the reference's default): a list of PIL images."""
        if output_type == "latent":
            return lat
        if output_type not in ("pil", "pt", "np"):
            raise ValueError(f"output_type {output_type!r}")
        if self.vae_decode is None:
            raise L.OmgHipError("no VAE attached: use output_type='latent' or construct the pipeline with "
                                "vae_decode=omg_amd.vae.AutoencoderKLDecoder(...).decode_latents")
        images = self.vae_decode(lat)
        if output_type == "pt":
            return images
        arr = images.detach().float().cpu().permute(0, 2, 3, 1).numpy()
        if output_type == "np":
            return arr
        from PIL import Image
        return [Image.fromarray(a) for a in (arr * 255).round().astype("uint8")]

    # ------------------------------------------------------------------ n independent requests in lock-step
    @torch.no_grad()
    def generate_many(self, requests: Sequence[dict], *, height: Optional[int] = None, width: Optional[int] = None,
                      num_inference_steps: int = 50, guidance_scale: float = 5.0, cross_attention_kwargs=None,
                      original_size=None, crops_coords_top_left=(0, 0), target_size=None, controller=None,
                      concept_models: Optional[ConceptModels] = None, stage: Optional[int] = None,
                      lora_list: Optional[Sequence[str]] = None, styleL: Optional[bool] = None, use_graph: bool = False,
                      trajectory: Optional[list] = None, fusion_start: int = FUSION_START, lora_mode: str = "merged",
                      controlnet=None, controlnet_image: Optional[torch.Tensor] = None, controlnet_conditioning_scale: float = 1.0,
                      identitynet=None, identitynet_conditioning_scale: float = 1.0, dedup: bool = False,
                      concept_lora: bool = True, concept_shard=None,
                      main_adapters: Optional[Sequence[Tuple[str, float]]] = None,
                      concept_adapters: Optional[Sequence[Tuple[str, float]]] = None, concept_adapter_scale: float = 1.0,
                      stage_cache: Optional[StageCache] = None, drop_unc0: bool = False, guess_mode: bool = False,
                      control_guidance_start=0.0, control_guidance_end=1.0, negative_original_size=None, negative_crops_coords_top_left=(0, 0),
                      negative_target_size=None, callback: Optional[Callable] = None, callback_steps: Optional[int] = None,
                      callback_on_step_end: Optional[Callable] = None, callback_on_step_end_tensor_inputs=("latents",)) -> torch.Tensor:
        """Each request: dict(prompt_embeds (2,77,Cx), negative_prompt_embeds, pooled_prompt_embeds (2,P),
        negative_pooled_prompt_embeds, region_prompt_embeds [(neg, pos, neg_pooled, pos_pooled)] * K, region_masks [K],
        latents | generator).  Returns final latents (n, 2, C, H/8, W/8): [base sample, edited sample] per request.
        All requests must agree on which concepts have a mask.

        ``controlnet`` + ``controlnet_image`` (1|n,3,H,W in [0,1]): ControlNet on the MAIN pass of every step
        (lora_pipeline.py:519-536; ``controlnet2``/``t2i_image`` of instantid_pipeline.py:574-592).
        Round 6: ``controlnet`` may be a LIST of ControlNets (or an object with ``.nets``: the reference wraps a list in diffusers' MultiControlNetModel,
        lora_pipeline.py:175-176) with a list of images and of scales — every net sees its own image at its own scale, the residuals are summed in list
        order (:366-383, :511-512); ``control_guidance_start`` / ``_end`` (scalars or one per net) give ``controlnet_keep[i]`` — the per-step factor 0 / 1 on
        the conditioning scale (:275-286, :421-428, :511-517; a net whose factor is 0 in a step is not run: adding its zero residuals = not adding them),
        for the IdentityNet and the t2i ControlNet of the InstantID flow too (instantid_pipeline.py:477-483, :566-578); ``guess_mode=True``
        (:497-503, :531-535): the nets see only the CONDITIONAL rows ``[cond0, cond1]`` of every request, with diffusers' logspace residual scaling, and the
        unconditional rows get no residual.
        ``negative_original_size`` + ``negative_target_size`` (+ ``negative_crops_coords_top_left``): SDXL's negative micro-conditioning as the reference
        EXECUTES it (lora_pipeline.py:459-474): ``cat([negative_ids, ids]).repeat(2, 1)`` — the four main rows get [negative, positive, negative,
        positive] time ids, i.e. alternating, NOT [neg, neg, pos, pos] like the prompt embeddings.  Kept as executed.
        ``callback_on_step_end(pipe, i, t, {name: tensor})`` after every step with the tensors named in ``callback_on_step_end_tensor_inputs``
        (``latents`` (2, C, H/8, W/8) fp32, ``prompt_embeds`` (4, 77, Cx) = [neg, neg, pos, pos], ``negative_prompt_embeds`` (2, 77, Cx)), once per request;
        a returned ``{"latents": ...}`` replaces the request's latents for the next step (:617-626).  Returned prompt embeddings are refused (the
        cached cross-attention K / V would have to follow).  ``callback(i, t, latents)`` every ``callback_steps`` steps (:629-632).  Callbacks switch off
        ``dedup`` / ``stage_cache`` (they may make the two samples differ) and run between hipGraph replays.
        ``identitynet`` (InstantID, instantid_pipeline.py:638-674): ControlNet on the CONCEPT pass fed with the face tokens and
        the request's ``kps_image`` (1,3,H,W); requests then also carry ``region_image_embeds`` = [(2,16,Cx) [zero-id, id]] * K and
        the UNet must have an :class:`omg_amd.ip_adapter.IPAdapter` installed.

        ``main_adapters`` / ``concept_adapters`` [(name, weight)]: adapters of ``concept_models.bank`` that are simply ACTIVE on the main /
        on every concept row — PEFT's state after ``load_lora_weights`` when nobody calls ``set_adapters`` (inference_instantid.py:220-222
        loads a style LoRA into both pipes and the InstantID loop never selects adapters).  Main rows run them at the caller's
        ``cross_attention_kwargs["scale"]`` (instantid_pipeline.py:596-616), concept rows at ``concept_adapter_scale`` (the concept UNet is
        called with ``cross_attention_kwargs=None``, :665-674: scale 1.0).  ``styleL=True`` is ``main_adapters=[("style", 1.0)]`` plus the
        LoRA flow's [concept, style] combination on the concept rows.

        ``stage_cache`` (:class:`StageCache`, OFF by default): the steps in front of the first fused step of a stage-2 call
        repeat the stage-1 call of the same image; with a cache shared by the two calls the second one starts at step ``fusion_start + 1``.
        ``drop_unc0=True`` (with a cache that also holds the stage-1 call's trajectory of the base sample): the resumed steps do not compute
        ``unc0`` — 3 main rows per request instead of 4; the base sample's latents come from the cache at every step (last item:
        100 + 238 = 338 of the reference's 536 sample-forwards for both stages; bitwise the same latents, tests/test_pipeline_gpu.py).

        ``dedup=True`` (OFF by default; SURVEY §7.4, "flag when used"): the reference duplicates the latents (:409) and is called
        with two equal prompts, so until the first fused step the two samples of a request are the same computation twice
        (stage 1: for the whole call).  With the batch-invariant kernels of this package those steps run ``[unc, cond]`` once per
        request and write the result to both samples — bitwise the same latents as the full batch (tests/test_pipeline_gpu.py),
        16 x 2 of the 336 sample-forwards of a 50-step stage-2 call fewer.  Used only where it is provably exact: equal prompt /
        pooled / negative embeddings of the two samples, no controller or a pure-replacement one, and a ControlNet conditioning image
        that is shared (1 image) or per request (n images) — with one image per main row the call falls back to the full batch.

        ``concept_shard`` (:class:`omg_amd.parallel.ConceptShard`, every rank of its group calls with the SAME requests): the step's
        independent forward units — the main block of a request, each concept pair — are split over the ranks, one all_gather
        per step exchanges the noise predictions and every rank applies the same fusion + CFG + scheduler kernel, so all ranks
        return the same latents, bitwise equal to the unsharded call (SURVEY §8(e) finer-grain option; latency, not throughput)."""
        dev, dt = self.unet.device, self.unet.dtype
        n = len(requests)
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        lora_list = list(lora_list or [])
        K = len(lora_list)
        S = num_inference_steps
        self.scheduler.set_timesteps(S, device=dev)
        ts = self.scheduler.timesteps.to(torch.float32)
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        Cl = self.unet.config.in_channels
        # ---- ControlNet plan: one net without guess_mode runs on the engine's original single-net path (twin / shard / stage-cache aware); a list of
        # nets or guess_mode on the general path `cn_multi` below (full batch only)
        cn_multi = None
        n_cn = 0
        if controlnet is not None:
            nets = list(controlnet.nets) if hasattr(controlnet, "nets") else list(controlnet) if isinstance(controlnet, (list, tuple)) else [controlnet]
            imgs = list(controlnet_image) if isinstance(controlnet_image, (list, tuple)) else [controlnet_image]
            scs = list(controlnet_conditioning_scale) if isinstance(controlnet_conditioning_scale, (list, tuple)) else [controlnet_conditioning_scale] * len(nets)
            if not nets or len(imgs) != len(nets) or len(scs) < len(nets) or any(im is None for im in imgs):
                raise ValueError(f"{len(nets)} ControlNet(s) need as many conditioning images and scales (got {len(imgs)} image(s), {len(scs)} scale(s))")
            n_cn = len(nets)
            if n_cn == 1 and not guess_mode:
                controlnet, controlnet_image, controlnet_conditioning_scale = nets[0], imgs[0], float(scs[0])
            else:
                cn_multi = SimpleNamespace(nets=nets, images=imgs, scales=[float(x) for x in scs[:n_cn]], guess=bool(guess_mode))
                controlnet, controlnet_image, controlnet_conditioning_scale = None, None, 1.0
        # controlnet_keep (lora_pipeline.py:421-428): per step, per net, 0.0 or 1.0
        cn_keep = controlnet_keep(S, control_guidance_start, control_guidance_end, max(1, n_cn))
        if len(cn_keep[0]) != max(1, n_cn):
            raise ValueError(f"control_guidance_start / _end must be scalars or hold one entry per ControlNet ({max(1, n_cn)})")
        # ---- per-request tensors
        lats, ehs_l, text_l, masks_l, cehs_l, ctext_l, ip_l, kps_l = [], [], [], [], [], [], [], []
        active = None
        for r in requests:
            pe, ne = r["prompt_embeds"], r["negative_prompt_embeds"]
            if pe.shape[0] != 2:
                raise ValueError("prompt_embeds must hold the 2 global prompts [p, p] (lora_pipeline.py:291)")
            lat0 = self.prepare_latents(1, Cl, height, width, torch.float32, dev, r.get("generator"), r.get("latents"))
            lats.append(torch.cat([lat0, lat0.clone()]))                                            # duplicated x2 (:409)
            ehs_l.append(torch.cat([ne, pe], dim=0).to(device=dev, dtype=dt))                       # CFG order [neg, pos] (:467-474)
            text_l.append(torch.cat([r["negative_pooled_prompt_embeds"], r["pooled_prompt_embeds"]], dim=0).to(device=dev, dtype=dt))
            rpe = r.get("region_prompt_embeds") or []
            if len(rpe) != K:
                raise ValueError("one (neg_embeds, pos_embeds, neg_pooled, pos_pooled) tuple per entry of lora_list is required")
            masks = [None] * K
            if stage == 2:
                rm = r.get("region_masks")
                if rm is None or len(rm) != K:
                    raise ValueError("stage 2 needs one region mask (or None) per entry of lora_list")
                masks = [m.to(device=dev, dtype=torch.float32).contiguous() if m is not None else None for m in rm]
            act = [c for c in range(K) if masks[c] is not None]
            if active is None:
                active = act
            elif act != active:
                raise ValueError("requests batched together must have masks for the same concepts")
            masks_l.append(masks)
            if act:
                cehs_l.append(torch.cat([torch.cat([rpe[c][0], rpe[c][1]], dim=0) for c in act], dim=0).to(device=dev, dtype=dt))
                ctext_l.append(torch.cat([torch.cat([rpe[c][2], rpe[c][3]], dim=0) for c in act], dim=0).to(device=dev, dtype=dt))
                if identitynet is not None:
                    rie = r.get("region_image_embeds")
                    if rie is None or len(rie) != K or r.get("kps_image") is None:
                        raise ValueError("identitynet needs region_image_embeds [(2,16,Cx)] * K and kps_image per request")
                    ip_l.append(torch.cat([rie[c] for c in act], dim=0).to(device=dev, dtype=dt))
                    kps_l.append(r["kps_image"].to(device=dev, dtype=torch.float32))
        Ka = len(active)
        shard = concept_shard if (concept_shard is not None and concept_shard.world > 1) else None
        has_cb = callback is not None or callback_on_step_end is not None
        if has_cb:
            if shard is not None:
                raise L.OmgHipError("callbacks with concept_shard are not built: every rank would have to run the same callback")
            bad = [k for k in callback_on_step_end_tensor_inputs if k not in _CALLBACK_TENSORS]
            if bad:
                raise ValueError(f"callback_on_step_end_tensor_inputs {bad}: only {list(_CALLBACK_TENSORS)} exist (lora_pipeline.py:165)")
            if callback is not None and (callback_steps is None or int(callback_steps) <= 0):
                raise ValueError("callback needs a positive callback_steps")
            dedup, stage_cache, drop_unc0 = False, None, False
        if cn_multi is not None:
            if shard is not None:
                raise L.OmgHipError("concept_shard with a list of ControlNets / guess_mode is not built: run the unsharded call")
            dedup, stage_cache, drop_unc0 = False, None, False        # the general ControlNet path runs the full batch
        twin = bool(dedup) and shard is None and (controller is None or getattr(controller, "is_pure_replacement", False))
        if twin and controlnet is not None and controlnet_image is not None and controlnet_image.shape[0] not in (1, n):
            twin = False      # one conditioning image per MAIN ROW (4n): the two samples of a request may differ, and the 2n-row twin batch
                              # would not divide it — full batch instead
        if twin:      # both samples of every request must be the same computation: [neg0, neg1, pos0, pos1] with equal halves.
            # ONE fused device comparison and one host sync per call (four torch.equal syncs per request before every call)
            e_all, t_all = torch.stack(ehs_l), torch.stack(text_l)                   # (n, 4, 77, Cx), (n, 4, P)
            same = (e_all[:, 0] == e_all[:, 1]).all() & (e_all[:, 2] == e_all[:, 3]).all() & (t_all[:, 0] == t_all[:, 1]).all() & (t_all[:, 2] == t_all[:, 3]).all()
            twin = bool(same)
        Hl, Wl = lats[0].shape[2:]
        fuse_possible = stage == 2 and Ka > 0 and S > fusion_start + 1
        nm, ncn = 4 * n, 2 * Ka * n                       # rows of the main block / the concept block
        nb = nm + (ncn if fuse_possible else 0)
        ehs = torch.cat(ehs_l, dim=0).contiguous()                                                  # (4n, 77, Cx)
        tids_main = self._add_time_ids(original_size, crops_coords_top_left, target_size, nm, dev)
        if negative_original_size is not None and negative_target_size is not None:
            # lora_pipeline.py:459-474 as executed: cat([negative_add_time_ids, add_time_ids]).repeat(batch_size, 1) -> rows [neg, pos, neg, pos]
            tids_neg = self._add_time_ids(negative_original_size, negative_crops_coords_top_left, negative_target_size, nm, dev)
            tids_main = torch.where((torch.arange(nm, device=dev) % 2 == 0)[:, None], tids_neg, tids_main)
        emb_main = self._all_step_embeddings(ts, torch.cat(text_l, dim=0), tids_main)
        slots: List[int] = []
        if fuse_possible and concept_models is None:
            raise ValueError("stage 2 needs concept_models")
        # ---- LoRA slots.  Concept passes: set_adapters(lora) or set_adapters([lora, "style"], [0.7, 0.5]) (lora_pipeline.py:588-591).
        # MAIN pass: inference_lora.py:162-164 loads the style LoRA into the main pipe as well, and the main UNet is called with
        # cross_attention_kwargs={"scale": 0.8} (:546-566), so with styleL every main sample runs with adapter "style" at PEFT
        # weight 1.0 x scale — one more slot of the same bank, selected for the main rows in plain AND fused steps.
        # LoRA scale: the concept UNet is always called with cross_attention_kwargs={'scale': 0.8} (hard-coded, :592-598); the main UNet
        # (style slot) with the caller's cross_attention_kwargs (:546-566; PEFT's default scale is 1.0)
        main_scale = float((cross_attention_kwargs or {}).get("scale", 1.0))
        # InstantID's concepts are identities (IP-Adapter tokens + IdentityNet), not LoRA adapters: its `lora_list` only counts them
        main_adapters = [(str(a), float(w)) for a, w in (main_adapters or ())] or ([("style", 1.0)] if styleL else [])
        concept_adapters = [(str(a), float(w)) for a, w in (concept_adapters or ())]
        if concept_lora and concept_adapters:
            raise ValueError("concept_adapters is the InstantID flow's 'whatever is active' rule; the LoRA flow selects adapters per concept (lora_list)")
        bank = concept_models.bank if (concept_models is not None and (concept_lora or concept_adapters or main_adapters)) else None
        combos = []
        if fuse_possible and concept_lora:
            combos = [((lora_list[c], 0.7), ("style", 0.5)) if styleL else ((lora_list[c], 1.0),) for c in active]
        scales = [CONCEPT_LORA_SCALE] * len(combos)
        if fuse_possible and concept_adapters:              # one combination for every concept row
            combos, scales = [tuple(concept_adapters)], [float(concept_adapter_scale)]
        main_slot = -1
        if main_adapters:
            if bank is None or any(a not in bank.adapters for a, _ in main_adapters):
                raise ValueError(f"adapters {[a for a, _ in main_adapters]} on the main pass need concept_models with a LoraBank holding them"
                                 + (" (styleL=True: the adapter named 'style')" if styleL else ""))
            main_slot = len(combos)
            combos = combos + [tuple(main_adapters)]
            scales.append(main_scale)
        if concept_adapters and (bank is None or any(a not in bank.adapters for a, _ in concept_adapters)):
            raise ValueError(f"concept_adapters {[a for a, _ in concept_adapters]} are not in concept_models' LoraBank")
        if bank is not None and combos:
            if [tuple(c) for c in combos] != list(bank.slots) or bank.scale != tuple(scales) or bank.mode != lora_mode:
                bank.build(combos, scale=scales, mode=lora_mode)
        if fuse_possible:
            if bank is None or not (concept_lora or concept_adapters):
                slots = [-1] * ncn
            elif concept_adapters:
                slots = [0] * ncn
            else:
                slots = [s for _ in range(n) for s in range(Ka) for _ in range(2)]
            c_ehs = torch.cat(cehs_l, dim=0).contiguous()                                           # (2Ka*n, 77, Cx)
            emb_conc = self._all_step_embeddings(ts, torch.cat(ctext_l, dim=0),
                                                 self._add_time_ids(original_size, crops_coords_top_left, target_size, ncn, dev),
                                                 unet=getattr(concept_models, "_unet", None))
        merged = bank is not None and bank.mode == "merged"
        # one batched forward needs ONE UNet behind both pipes (omg_amd.compat shares it; hand-built ConceptModels may not)
        c_unet = getattr(concept_models, "_unet", self.unet) if concept_models is not None else self.unet
        shared = c_unet is self.unet
        batched = fuse_possible and (bank is None or merged) and shared
        if use_graph and not (bank is None or merged) and (fuse_possible or main_slot >= 0):
            raise L.OmgHipError("use_graph needs lora_mode='merged' (segment-mode K/V projections are not pointer-stable)")
        use_cn = controlnet is not None
        use_idn = identitynet is not None and fuse_possible
        if use_cn and controlnet_image is None:
            raise ValueError("controlnet needs controlnet_image")
        if use_idn and not batched:
            raise L.OmgHipError("identitynet runs in the batched mode only: merged / no LoRA, and concept_models built on the pipeline's own "
                                "UNet (omg_amd.compat's from_pretrained shares it between the main and the concept pipe)")
        if shard is not None and ((fuse_possible and not batched) or use_idn or (controller is not None and not getattr(controller, "is_pure_replacement", False))):
            raise L.OmgHipError("concept_shard needs the batched step (merged / no LoRA on one shared UNet), a pure-replacement controller and no IdentityNet")
        mshape = tuple(masks_l[0][active[0]].shape) if active else (0, 0)
        D = emb_main.shape[-1]
        # ---- stage cache: everything that determines the latents entering step fusion_start + 1 of request j; decided in
        # front of the engine because a resumed call without `unc0` runs a different row plan
        first = 0
        cache_keys: List[str] = []
        cache_hit = None
        base_traj = None                      # drop_unc0: per step k the (n, C, H, W) latents / model inputs of the base samples
        if use_cn and controlnet_image.shape[0] not in (1, n):
            # one conditioning image per MAIN ROW (4n): the cache key and drop_unc0's compact rows address images per REQUEST — not built for
            # that layout (the full batch below handles it); neither is used instead of being keyed / indexed wrongly
            stage_cache, drop_unc0 = None, False
        if stage_cache is not None and shard is None and S > fusion_start + 1:
            common = (S, float(guidance_scale), type(self.scheduler).__name__, fusion_start, height, width, tuple(original_size),
                      None if negative_original_size is None or negative_target_size is None else
                      (tuple(negative_original_size), tuple(negative_crops_coords_top_left), tuple(negative_target_size)),
                      tuple(crops_coords_top_left), tuple(target_size), str(dt), tuple(main_adapters), main_scale, lora_mode if main_adapters else None,
                      id(self.unet), getattr(self.unet, "weights_version", 0), getattr(bank, "version", None) if main_adapters else None,
                      None if controller is None else (type(controller).__name__, getattr(controller, "is_pure_replacement", False),
                                                       getattr(controller, "num_self_replace", None)),
                      None if not use_cn else (id(controlnet), getattr(controlnet, "weights_version", 0), float(controlnet_conditioning_scale),
                                               tuple(k[0] for k in cn_keep[: fusion_start + 1])))
            coef_key = self.scheduler.coef_table(dev)
            for j in range(n):
                cn_img = None if not use_cn else (controlnet_image if controlnet_image.shape[0] == 1 else controlnet_image[j: j + 1])
                cache_keys.append(StageCache.digest(common, coef_key, lats[j], ehs_l[j], text_l[j], cn_img))
            hit = [stage_cache.get(k) for k in cache_keys]
            if fuse_possible and all(h is not None for h in hit):
                first, cache_hit = fusion_start + 1, hit                 # every request resumes: the plain steps are not run at all
                stage_cache.hits += n
                if drop_unc0 and batched and not use_idn and (controller is None or getattr(controller, "is_pure_replacement", False)):
                    bases = [stage_cache.get_base(k, first, S) for k in cache_keys]
                    if all(b is not None for b in bases):
                        base_traj = {k: (torch.cat([b[k][0] for b in bases]).to(dev), torch.cat([b[k][1] for b in bases]).to(device=dev, dtype=dt))
                                     for k in range(first + 1, S + 1)}
            else:
                stage_cache.misses += n
        drop0 = base_traj is not None
        if drop0:      # the compact-row machinery of the concept shard, on ONE rank: main units of three rows [unc1, cond0, cond1]
            from . import parallel as _par0
            shard = _par0.ConceptShard(rank=0, world=1)
            twin = False
        # ---- persistent engine state (static buffers + captured step graphs), reused across calls of the same shape
        key = (n, S, Cl, Hl, Wl, K, tuple(active), fuse_possible, fusion_start, type(self.scheduler).__name__, float(guidance_scale),
               str(dt), batched, bool(styleL), tuple(main_adapters), tuple(concept_adapters), float(concept_adapter_scale), main_scale,
               tuple(lora_list), mshape, tuple(ehs.shape), id(controller), id(controlnet), id(identitynet),
               float(controlnet_conditioning_scale), float(identitynet_conditioning_scale), twin,
               (shard.rank, shard.world) if shard is not None else None, drop0,
               None if cn_multi is None else (tuple(id(x) for x in cn_multi.nets), tuple(tuple(im.shape) for im in cn_multi.images), cn_multi.guess))
        eng = self._engines.pop(key, None)
        if eng is not None:
            self._engines[key] = eng                       # most recently used last
        if eng is None:
            eng = SimpleNamespace(graphs={}, warmed=set(), pool=None, coef=None)
            eng.lat = torch.empty((2 * n, Cl, Hl, Wl), dtype=torch.float32, device=dev)
            eng.xin = torch.empty((nb, Cl, Hl, Wl), dtype=dt, device=dev)
            # drop0: the rows of `unc0` are never written — zeros, so that the step kernel's (discarded) update of the base sample stays finite
            eng.nout = (torch.zeros if drop0 else torch.empty)((nb, Cl, Hl, Wl), dtype=torch.float32, device=dev)
            eng.step_idx = torch.zeros(1, dtype=torch.int32, device=dev)
            eng.ehs = torch.empty_like(ehs)
            eng.emb_main = torch.empty((S, nm, D), dtype=dt, device=dev)
            eng.emb_cur_main = torch.empty((nm, D), dtype=dt, device=dev)
            eng.masks = [[torch.empty(mshape, dtype=torch.float32, device=dev) if c in active else None for c in range(K)] for _ in range(n)]
            if fuse_possible:
                eng.c_ehs = torch.empty_like(c_ehs)
                eng.emb_conc = torch.empty((S, ncn, D), dtype=dt, device=dev)
                eng.emb_cur_conc = torch.empty((ncn, D), dtype=dt, device=dev)
            if batched:
                eng.ehs_all = torch.empty((nb,) + tuple(ehs.shape[1:]), dtype=dt, device=dev)
                eng.emb_all = torch.empty((S, nb, D), dtype=dt, device=dev)
                eng.emb_cur_all = torch.empty((nb, D), dtype=dt, device=dev)
            if twin:      # compact rows [unc, cond] per request for the steps in which the two samples coincide
                eng.xin2 = torch.empty((2 * n, Cl, Hl, Wl), dtype=dt, device=dev)
                eng.nout2 = torch.empty((2 * n, Cl, Hl, Wl), dtype=torch.float32, device=dev)
                eng.ehs2 = torch.empty((2 * n,) + tuple(ehs.shape[1:]), dtype=dt, device=dev)
                eng.emb_main2 = torch.empty((S, 2 * n, D), dtype=dt, device=dev)
                eng.emb_cur2 = torch.empty((2 * n, D), dtype=dt, device=dev)
                if use_cn:
                    eng.cn_emb2 = torch.empty((S, 2 * n, D), dtype=dt, device=dev)
                    eng.cn_emb_cur2 = torch.empty((2 * n, D), dtype=dt, device=dev)
            if use_cn:
                eng.cn_image = torch.empty((controlnet_image.shape[0], 3, height, width), dtype=torch.float32, device=dev)
                eng.cn_emb = torch.empty((S, nm, D), dtype=dt, device=dev)
                eng.cn_emb_cur = torch.empty((nm, D), dtype=dt, device=dev)
            if cn_multi is not None:
                rows_cn = 2 * n if cn_multi.guess else nm
                eng.cnm_image = [torch.empty((im.shape[0], 3, height, width), dtype=torch.float32, device=dev) for im in cn_multi.images]
                eng.cnm_emb = [torch.empty((S, rows_cn, D), dtype=dt, device=dev) for _ in cn_multi.nets]
                eng.cnm_emb_cur = [torch.empty((rows_cn, D), dtype=dt, device=dev) for _ in cn_multi.nets]
                if cn_multi.guess:      # the conditional rows [cond0, cond1] of every request, compact
                    eng.cnm_x = torch.empty((rows_cn, Cl, Hl, Wl), dtype=dt, device=dev)
                    eng.cnm_ehs = torch.empty((rows_cn,) + tuple(ehs.shape[1:]), dtype=dt, device=dev)
            if use_idn:
                eng.ip_all = torch.empty((ncn,) + tuple(ip_l[0].shape[1:]), dtype=dt, device=dev)
                eng.kps_all = torch.empty((n, 3, height, width), dtype=torch.float32, device=dev)
                eng.idn_emb = torch.empty((S, ncn, D), dtype=dt, device=dev)
                eng.idn_emb_cur = torch.empty((ncn, D), dtype=dt, device=dev)
            if shard is not None:      # per regime (plain / fused): this rank's units, its compact local batch and every rank's row map
                from . import parallel as _par
                eng.sh = {}
                for fz in ((True,) if drop0 else (False, True) if fuse_possible else (False,)):
                    if drop0:      # every unit on this one rank; a main unit is rows 1..3 of the request's block (`unc0` is not computed)
                        per_rank = [(list(range(n)), [(j, c) for j in range(n) for c in range(Ka)])]
                        mrows = [4 * j + r for j in range(n) for r in (1, 2, 3)]
                        rows = [(mrows + [4 * j + 3 for j, _ in per_rank[0][1] for _ in range(2)],
                                 mrows + [nm + 2 * Ka * j + 2 * c + r for j, c in per_rank[0][1] for r in range(2)])]
                    else:
                        per_rank = _par.assign_units(n, Ka, fz, shard.world)
                        rows = _par.unit_rows(per_rank, n, Ka)
                    mains, concs = per_rank[shard.rank]
                    src, dst = rows[shard.rank]
                    sh = SimpleNamespace(mains=mains, concs=concs, rows=len(src), n_main=(3 if drop0 else 4) * len(mains), counts=[len(r[1]) for r in rows])
                    sh.src = torch.tensor(src, dtype=torch.long, device=dev)
                    sh.dsts = [torch.tensor(r[1], dtype=torch.long, device=dev) for r in rows]
                    sh.x = torch.empty((sh.rows, Cl, Hl, Wl), dtype=dt, device=dev)
                    sh.y = torch.zeros((max(sh.counts), Cl, Hl, Wl), dtype=torch.float32, device=dev)
                    sh.ehs = torch.empty((sh.rows,) + tuple(ehs.shape[1:]), dtype=dt, device=dev)
                    sh.emb = torch.empty((S, sh.rows, D), dtype=dt, device=dev)
                    sh.emb_cur = torch.empty((sh.rows, D), dtype=dt, device=dev)
                    if use_cn and sh.n_main:
                        sh.cn_emb = torch.empty((S, sh.n_main, D), dtype=dt, device=dev)
                        sh.cn_emb_cur = torch.empty((sh.n_main, D), dtype=dt, device=dev)
                        sh.cn_image = torch.empty((len(mains) if controlnet_image.shape[0] != 1 else 1, 3, height, width), dtype=torch.float32, device=dev)
                    eng.sh[fz] = sh
            while len(self._engines) >= self.max_engines:      # least recently used first; its captured graphs go with it
                old_key = next(iter(self._engines))
                self._engines.pop(old_key)
                self.engine_evictions += 1
                if self.engine_evictions in (1, 10, 100):
                    import warnings
                    warnings.warn(f"LoraMultiConceptPipeline: step engine evicted ({self.engine_evictions} so far): more than max_engines="
                                  f"{self.max_engines} call shapes alternate and each eviction re-captures its hipGraphs; raise pipe.max_engines")
            self._engines[key] = eng
            # the key holds id()s: keep the objects alive so that an id cannot be recycled for a different object
            eng.refs = (controller, controlnet, identitynet, concept_models, cn_multi.nets if cn_multi is not None else None)
            eng.epoch = pointer_epoch()
        # ---- load this call's inputs into the static buffers (device-to-device copies; graphs keep their pointers)
        lat = eng.lat
        lat.copy_(torch.cat(lats, dim=0))
        eng.ehs.copy_(ehs)
        eng.emb_main.copy_(emb_main)
        coef_now = self.scheduler.coef_table(dev)          # per call: the key holds the scheduler's class and step count, not its
        if eng.coef is None:                               # configuration — a re-configured scheduler must not meet a cached table
            eng.coef = coef_now.clone()                    # the engine's OWN buffer: the scheduler's cached table is shared and read-only
        else:
            eng.coef.copy_(coef_now)                       # in place: captured graphs keep the pointer
        eng.step_idx.zero_()
        for j in range(n):
            for c in active:
                eng.masks[j][c].copy_(masks_l[j][c])
        xin, nout, step_idx, coef = eng.xin, eng.nout, eng.step_idx, eng.coef
        if cache_hit is not None:                                        # resume (decided in front of the engine, above)
            lat.copy_(torch.cat([h.to(dev) for h in cache_hit], dim=0))
            step_idx.fill_(first)
            if controller is not None:
                controller.cur_step = first                              # the host-side counters the skipped steps would have ticked
        cin0 = self.scheduler.cin0(dev) if first == 0 else torch.tensor([self.scheduler.cin[first]], dtype=torch.float32, device=dev)
        for j in range(n):
            ops.scale_model_input(lat[2 * j: 2 * j + 2], cin0, xin[4 * j: 4 * j + 4])
        main_kw = dict(cross_attention_kwargs or {})
        main_kw.pop("scale", None)
        main_kw["omg_main_batch"] = 4
        main_kw["omg_images"] = n
        if fuse_possible:
            eng.c_ehs.copy_(c_ehs)
            eng.emb_conc.copy_(emb_conc)
        if batched:
            eng.ehs_all[:nm].copy_(ehs)
            eng.ehs_all[nm:].copy_(c_ehs)
            eng.emb_all[:, :nm].copy_(emb_main)
            eng.emb_all[:, nm:].copy_(emb_conc)
            state_all = concept_models.lora_state([main_slot + 1] * nm + [s + 1 for s in slots], merged=True) if bank is not None else None
        # main-only forwards (plain steps; the main half of un-batched fused steps): base weights, or the style slot
        state_main = None
        if main_slot >= 0:
            state_main = concept_models.lora_state([main_slot + 1] * nm, merged=True) if merged else concept_models.lora_state([main_slot] * nm, merged=False)
        tids_m = tids_main
        if use_cn:
            eng.cn_image.copy_(controlnet_image.to(device=dev, dtype=torch.float32))
            t_all = ts.reshape(S, 1).expand(S, nm).reshape(-1).contiguous()
            eng.cn_emb.copy_(controlnet.time_embed(t_all, S * nm, torch.cat(text_l, dim=0).repeat(S, 1), tids_m.repeat(S, 1)).view(S, nm, D))
        if cn_multi is not None:
            cond_rows = torch.tensor([4 * j + r for j in range(n) for r in (2, 3)], dtype=torch.long, device=dev)
            text_cn, tids_cn, rows_cn = torch.cat(text_l, dim=0), tids_m, nm
            if cn_multi.guess:      # add_text_embeds.chunk(2)[1] / add_time_ids.chunk(2)[1] / prompt_embeds.chunk(2)[1] (lora_pipeline.py:497-503)
                text_cn, tids_cn, rows_cn = text_cn.index_select(0, cond_rows), tids_m.index_select(0, cond_rows), 2 * n
                eng.cnm_ehs.copy_(ehs.index_select(0, cond_rows))
            t_all = ts.reshape(S, 1).expand(S, rows_cn).reshape(-1).contiguous()
            for k_, net in enumerate(cn_multi.nets):
                eng.cnm_image[k_].copy_(cn_multi.images[k_].to(device=dev, dtype=torch.float32))
                eng.cnm_emb[k_].copy_(net.time_embed(t_all, S * rows_cn, text_cn.repeat(S, 1), tids_cn.repeat(S, 1)).view(S, rows_cn, D))
                if use_graph:
                    net.refresh_cross_kv(eng.cnm_ehs if cn_multi.guess else eng.ehs)
                    net.cond_features(eng.cnm_image[k_])
        if use_idn:
            eng.ip_all.copy_(torch.cat(ip_l, dim=0))
            eng.kps_all.copy_(torch.cat(kps_l, dim=0))
            t_all = ts.reshape(S, 1).expand(S, ncn).reshape(-1).contiguous()
            tids_c = self._add_time_ids(original_size, crops_coords_top_left, target_size, ncn, dev)
            eng.idn_emb.copy_(identitynet.time_embed(t_all, S * ncn, torch.cat(ctext_l, dim=0).repeat(S, 1), tids_c.repeat(S, 1)).view(S, ncn, D))
        state_twin = None
        if twin:
            rows2 = torch.tensor([4 * j + r for j in range(n) for r in (0, 2)], dtype=torch.long, device=dev)      # unc0, cond0 of each request
            eng.ehs2.copy_(ehs.index_select(0, rows2))
            eng.emb_main2.copy_(emb_main.index_select(1, rows2))
            if use_cn:
                eng.cn_emb2.copy_(eng.cn_emb.index_select(1, rows2))
            if main_slot >= 0:
                state_twin = concept_models.lora_state([main_slot + 1] * (2 * n), merged=True) if merged else concept_models.lora_state([main_slot] * (2 * n), merged=False)
        if shard is not None:
            for fz, sh in eng.sh.items():
                if not sh.rows:
                    continue
                src_e, src_m = (eng.ehs_all, eng.emb_all) if fz else (eng.ehs, eng.emb_main)
                sh.ehs.copy_(src_e.index_select(0, sh.dsts[shard.rank]))
                sh.emb.copy_(src_m.index_select(1, sh.dsts[shard.rank]))
                sh.state = None
                if bank is not None and (main_slot >= 0 or sh.concs):
                    sh.state = concept_models.lora_state([main_slot + 1] * sh.n_main + [jj + 1 for _, jj in sh.concs for _ in range(2)], merged=True)
                if use_cn and sh.n_main:
                    sh.cn_emb.copy_(eng.cn_emb.index_select(1, sh.dsts[shard.rank][: sh.n_main]))
                    sh.cn_image.copy_(eng.cn_image if eng.cn_image.shape[0] == 1 else eng.cn_image[sh.mains])
                if use_graph:
                    self.unet.refresh_cross_kv(sh.ehs, sh.state)
                    if use_cn and sh.n_main:
                        controlnet.refresh_cross_kv(sh.ehs[: sh.n_main])
                        controlnet.cond_features(sh.cn_image)
        if use_graph and shard is None:
            # cached cross-attention K/V (and ControlNet conditioning features) must be refreshed eagerly:
            # replayed graphs read the stored tensors
            if twin:
                self.unet.refresh_cross_kv(eng.ehs2, state_twin)
                if use_cn:
                    controlnet.refresh_cross_kv(eng.ehs2)
            self.unet.refresh_cross_kv(eng.ehs, state_main)
            if batched:
                self.unet.refresh_cross_kv(eng.ehs_all, state_all)
            if use_cn:
                controlnet.refresh_cross_kv(eng.ehs)
                controlnet.cond_features(eng.cn_image)
            if use_idn:
                identitynet.refresh_cross_kv(eng.ip_all)
                identitynet.cond_features(eng.kps_all)
                self.unet.refresh_ip_kv(eng.ip_all)      # the UNet's own image-prompt K / V^T of the concept samples

        # the conditioning scales of the CURRENT step (host floats baked into the zero convolutions' epilogues, hence part of the graph regime):
        # scale * controlnet_keep[i] (lora_pipeline.py:511-517; instantid_pipeline.py:566-578: the one window scales the IdentityNet and the t2i net)
        cur = SimpleNamespace(cn=float(controlnet_conditioning_scale), idn=float(identitynet_conditioning_scale), multi=())

        def set_step_scales(i: int) -> tuple:
            k0 = cn_keep[i][0]
            cur.cn, cur.idn = float(controlnet_conditioning_scale) * k0, float(identitynet_conditioning_scale) * k0
            if cn_multi is not None:
                cur.multi = tuple(sc * cn_keep[i][k_] for k_, sc in enumerate(cn_multi.scales))
            return (cur.cn if use_cn else None, cur.idn if use_idn else None, cur.multi)

        def multi_residuals():
            """The general ControlNet block of a step: every net of the list on the main rows (guess_mode: on the conditional rows only), residuals
            summed in list order like MultiControlNetModel.forward — first the nets' residuals among themselves, then onto the skips — and, in
            guess_mode, no residual for the unconditional rows (lora_pipeline.py:531-535 concatenates zeros)."""
            sum_d = sum_m = None
            if cn_multi.guess:
                eng.cnm_x.view(n, 2, Cl, Hl, Wl).copy_(xin[:nm].view(n, 4, Cl, Hl, Wl)[:, 2:4])
            for k_, net in enumerate(cn_multi.nets):
                sc = cur.multi[k_]
                if sc == 0.0:
                    continue
                ops.gather_step(eng.cnm_emb[k_], step_idx, eng.cnm_emb_cur[k_])
                d_, m_ = net(eng.cnm_x if cn_multi.guess else xin[:nm], None, encoder_hidden_states=eng.cnm_ehs if cn_multi.guess else eng.ehs,
                             controlnet_cond=eng.cnm_image[k_], conditioning_scale=sc, emb=eng.cnm_emb_cur[k_], guess_mode=cn_multi.guess)
                if sum_d is None:
                    sum_d, sum_m = d_, m_
                else:
                    for a_, b_ in zip(sum_d + [sum_m], d_ + [m_]):
                        ops.add_(a_.permute(0, 2, 3, 1), b_.permute(0, 2, 3, 1))
            if sum_d is None:
                return []
            if not cn_multi.guess:
                return [(0, nm, sum_d, sum_m)]
            return [(4 * j + 2, 4 * j + 4, [d_[2 * j: 2 * j + 2] for d_ in sum_d], sum_m[2 * j: 2 * j + 2]) for j in range(n)]

        def region_rows(j):
            return nm + 2 * Ka * j

        def fill_region_inputs():
            for j in range(n):                            # latent_model_input[3:4] duplicated (:583-585), per request
                r0 = region_rows(j)
                xin[r0: r0 + 2 * Ka].copy_(xin[4 * j + 3: 4 * j + 4].expand(2 * Ka, -1, -1, -1))

        def twin_body():
            """A step in which samples 0 and 1 of every request are still identical: one [unc, cond] forward per request, the noise
            prediction written to both samples, then the ordinary CFG + scheduler step on all four rows."""
            kw = dict(main_kw)
            kw["omg_twin"] = True
            x2, y2 = eng.xin2, eng.nout2
            x2.view(n, 2, Cl, Hl, Wl).copy_(xin[:nm].view(n, 2, 2, Cl, Hl, Wl)[:, :, 0])
            if use_cn and cur.cn != 0.0:
                ops.gather_step(eng.cn_emb2, step_idx, eng.cn_emb_cur2)
                d_, m_ = controlnet(x2, None, encoder_hidden_states=eng.ehs2, controlnet_cond=eng.cn_image,
                                    conditioning_scale=cur.cn, emb=eng.cn_emb_cur2)
                kw["omg_residuals"] = [(0, 2 * n, d_, m_)]
            ops.gather_step(eng.emb_main2, step_idx, eng.emb_cur2)
            self.unet.set_lora_state(state_twin)
            try:
                self.unet(x2, None, encoder_hidden_states=eng.ehs2, cross_attention_kwargs=kw, emb=eng.emb_cur2, out=y2)
            finally:
                self.unet.set_lora_state(None)
            nout[:nm].view(n, 2, 2, Cl, Hl, Wl).copy_(y2.view(n, 2, 1, Cl, Hl, Wl).expand(n, 2, 2, Cl, Hl, Wl))
            for j in range(n):
                ops.fuse_cfg_step(nout[4 * j: 4 * j + 4], lat[2 * j: 2 * j + 2], coef, step_idx, guidance_scale=guidance_scale,
                                  fuse=False, region_preds=[None] * K, masks=[None] * K,
                                  model_input_next=xin[4 * j: 4 * j + 4], advance=(j == n - 1))

        def step_body(fused: bool, twin_step: bool = False):
            """One denoising iteration; every per-step quantity is selected by the DEVICE step counter."""
            if twin_step:
                return twin_body()
            kw = dict(main_kw)
            residuals = []
            if use_cn and cur.cn != 0.0:                  # ControlNet on the main samples (lora_pipeline.py:519-536); scale 0 = outside its guidance window
                ops.gather_step(eng.cn_emb, step_idx, eng.cn_emb_cur)
                d_, m_ = controlnet(xin[:nm], None, encoder_hidden_states=eng.ehs, controlnet_cond=eng.cn_image,
                                    conditioning_scale=cur.cn, emb=eng.cn_emb_cur)
                residuals.append((0, nm, d_, m_))
            if cn_multi is not None:
                residuals.extend(multi_residuals())
            if fused and batched:
                fill_region_inputs()
                if use_idn:                               # IdentityNet on the concept samples (instantid_pipeline.py:638-648)
                    if cur.idn != 0.0:
                        ops.gather_step(eng.idn_emb, step_idx, eng.idn_emb_cur)
                        d_, m_ = identitynet(xin[nm:], None, encoder_hidden_states=eng.ip_all, controlnet_cond=eng.kps_all,
                                             conditioning_scale=cur.idn, emb=eng.idn_emb_cur)
                        residuals.append((nm, nb, d_, m_))
                    kw["omg_ip_tokens"], kw["omg_ip_rows"] = eng.ip_all, nm
                if residuals:
                    kw["omg_residuals"] = residuals
                ops.gather_step(eng.emb_all, step_idx, eng.emb_cur_all)
                self.unet.set_lora_state(state_all)
                try:
                    self.unet(xin, None, encoder_hidden_states=eng.ehs_all, cross_attention_kwargs=kw, emb=eng.emb_cur_all, out=nout)
                finally:
                    self.unet.set_lora_state(None)
            else:
                if residuals:
                    kw["omg_residuals"] = residuals
                ops.gather_step(eng.emb_main, step_idx, eng.emb_cur_main)
                self.unet.set_lora_state(state_main)
                try:
                    self.unet(xin[:nm], None, encoder_hidden_states=eng.ehs, cross_attention_kwargs=kw, emb=eng.emb_cur_main, out=nout[:nm])
                finally:
                    self.unet.set_lora_state(None)
                if fused:
                    fill_region_inputs()
                    ops.gather_step(eng.emb_conc, step_idx, eng.emb_cur_conc)
                    concept_models.unet_batched(xin[nm:], None, eng.c_ehs, slots, emb=eng.emb_cur_conc, out=nout[nm:])
            finish(fused)

        def finish(fused: bool):
            """Region fusion + CFG + scheduler update + next model input of every request (one launch each)."""
            for j in range(n):
                regs: List[Optional[torch.Tensor]] = [None] * K
                if fused:
                    for jj, c in enumerate(active):
                        r0 = region_rows(j) + 2 * jj
                        regs[c] = nout[r0: r0 + 2]
                ops.fuse_cfg_step(nout[4 * j: 4 * j + 4], lat[2 * j: 2 * j + 2], coef, step_idx, guidance_scale=guidance_scale,
                                  fuse=fused, region_preds=regs, masks=eng.masks[j] if fused else [None] * K,
                                  model_input_next=xin[4 * j: 4 * j + 4], advance=(j == n - 1))

        def shard_forward(fused: bool, _tw: bool = False):
            """This rank's units of the step: main blocks first, then concept pairs, as one compact batch."""
            sh = eng.sh[fused]
            kw = dict(main_kw)
            kw["omg_images"] = len(sh.mains)
            if drop0:
                kw["omg_main_batch"] = 3                  # [unc1, cond0, cond1]: cond1 borrows Q, K of cond0 = row 1 of the block
            sh.x.copy_(xin[:nm].index_select(0, sh.src))
            if use_cn and sh.n_main and cur.cn != 0.0:
                ops.gather_step(sh.cn_emb, step_idx, sh.cn_emb_cur)
                d_, m_ = controlnet(sh.x[: sh.n_main], None, encoder_hidden_states=sh.ehs[: sh.n_main], controlnet_cond=sh.cn_image,
                                    conditioning_scale=cur.cn, emb=sh.cn_emb_cur)
                kw["omg_residuals"] = [(0, sh.n_main, d_, m_)]
            ops.gather_step(sh.emb, step_idx, sh.emb_cur)
            self.unet.set_lora_state(sh.state)
            try:
                self.unet(sh.x, None, encoder_hidden_states=sh.ehs, cross_attention_kwargs=kw, emb=sh.emb_cur, out=sh.y[: sh.rows])
            finally:
                self.unet.set_lora_state(None)

        def run_step(i: int):
            fused = fuse_possible and i > fusion_start
            tw = twin and not fused          # the samples part ways at the first fused step (stage 1: never)
            scales_i = set_step_scales(i)
            if shard is not None:
                sh = eng.sh[fused]
                if sh.rows:
                    run_forward(shard_forward, fused, False, ("shard", fused, scales_i))
                elif controller is not None:
                    controller.cur_step += 1          # no unit of this step is ours: only the host-side step counter moves
                shard.exchange(sh.y, sh.counts, sh.dsts, nout)
                finish(fused)
                if drop0:      # the base samples were not computed: their latents and next model inputs are the stage-1 call's
                    bl, bx = base_traj[i + 1]
                    lat.view(n, 2, Cl, Hl, Wl)[:, 0].copy_(bl)
                    if i + 1 < S:
                        xin[:nm].view(n, 4, Cl, Hl, Wl)[:, 2].copy_(bx)
                return
            run_forward(step_body, fused, tw, (fused, scales_i))

        def run_forward(body, fused: bool, tw: bool, tag: tuple):
            if not use_graph:
                body(fused, tw)
                return
            if eng.epoch != pointer_epoch():          # a weight image or cached K/V the graphs point at was freed or re-allocated
                eng.graphs.clear()
                eng.warmed.clear()
                eng.pool = None                       # the pool dies with its last graph: a later capture must open a new one
                eng.epoch = pointer_epoch()
            win = controller._self_window() if controller is not None and hasattr(controller, "_self_window") else None
            regime = tag + (win, tw)
            g = eng.graphs.get(regime)
            if g is not None:
                g.replay()
                if controller is not None:
                    controller.cur_step += 1          # the replayed graph does not run the host-side counters
                return
            if regime not in eng.warmed:              # first step of a regime runs eagerly (lazy inits, caches)
                body(fused, tw)
                eng.warmed.add(regime)
                return
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, pool=eng.pool):
                body(fused, tw)                       # records the step; host counters tick as in eager mode
            if eng.pool is None:
                eng.pool = g.pool()
            eng.graphs[regime] = g
            g.replay()

        # ---- 8. denoising loop
        for i in range(first, S):
            run_step(i)
            if has_cb:
                t_i = self.scheduler.timesteps[i]
                for j in range(n):
                    lj = lat[2 * j: 2 * j + 2]
                    if callback_on_step_end is not None:
                        avail = {"latents": lj, "prompt_embeds": eng.ehs[4 * j: 4 * j + 4], "negative_prompt_embeds": eng.ehs[4 * j: 4 * j + 2]}
                        outs = dict(callback_on_step_end(self, i, t_i, {k: avail[k] for k in callback_on_step_end_tensor_inputs}) or {})
                        new = outs.pop("latents", lj)
                        for k_, v_ in outs.items():
                            if k_ in avail and v_ is not avail[k_] and not torch.equal(v_.to(avail[k_]), avail[k_]):
                                raise L.OmgHipError(f"callback_on_step_end returned a new {k_}: only `latents` can be replaced mid-loop")
                        if new is not lj:
                            lj.copy_(new.to(device=dev, dtype=torch.float32))
                            if i + 1 < S:      # the step kernel wrote the next model input from ITS latents: redo it from the callback's
                                ops.scale_model_input(lj, torch.tensor([self.scheduler.cin[i + 1]], dtype=torch.float32, device=dev), xin[4 * j: 4 * j + 4])
                    if callback is not None and i % int(callback_steps) == 0:
                        callback(i // getattr(self.scheduler, "order", 1), t_i, lj)
            if trajectory is not None:
                trajectory.append(lat.clone().view(n, 2, Cl, Hl, Wl))
            if cache_keys and first == 0 and i == fusion_start:         # the latents entering the first fused step of a stage-2 call
                for j, k in enumerate(cache_keys):
                    stage_cache.put(k, lat[2 * j: 2 * j + 2])
            if cache_keys and first == 0 and i >= fusion_start:         # ... and the base sample's way from there on (drop_unc0)
                for j, k in enumerate(cache_keys):
                    stage_cache.put_base(k, i + 1, lat[2 * j: 2 * j + 1], xin[4 * j + 2: 4 * j + 3])
        return lat.clone().view(n, 2, Cl, Hl, Wl)


class InstantidMultiConceptPipeline(LoraMultiConceptPipeline):
    """OMG + InstantID (/root/reference src/pipelines/instantid_pipeline.py:157-768, BASELINE config 3) on the same step engine.

    Reference roles -> here: ``self.controlnet`` (IdentityNet, applied to every CONCEPT pass with the face tokens as context and the
    key-point image as conditioning, :638-648) = ``identitynet``; ``self.controlnet2`` + ``t2i_image`` (optional pose/depth ControlNet on the
    MAIN pass, :574-616) = ``controlnet2``; the concept UNet's IP-Adapter processors (instantid_single_pieline.py:186-213) = an
    :class:`omg_amd.ip_adapter.IPAdapter` installed on the shared UNet.  Face detection / ArcFace embedding (``face_app``) and the
    Resampler that turns a 512-d embedding into 16 tokens run once per image outside the hot path: pass the
    resulting tokens as ``region_image_embeds`` = [(2, 16, Cx) = [tokens of the zero embedding, tokens of the identity]] per concept
    (:class:`omg_amd.resampler.Resampler` computes them from the embedding: ``resampler(torch.stack([zeros, emb]).view(2, 1, 512))``,
    instantid_single_pieline.py:221-243)."""

    def __init__(self, unet, identitynet, scheduler=None, controlnet2=None, **kw):
        super().__init__(unet, scheduler, **kw)
        self.controlnet = identitynet
        self.controlnet2 = controlnet2

    @torch.no_grad()
    def __call__(self, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                 image: Optional[torch.Tensor] = None, t2i_image: Optional[torch.Tensor] = None, height=None, width=None,
                 num_inference_steps: int = 50, guidance_scale: float = 5.0, generator=None, latents=None,
                 controlnet_conditioning_scale: float = 1.0, t2i_controlnet_conditioning_scale: float = 1.0, controller=None,
                 concept_models: Optional[ConceptModels] = None, stage: Optional[int] = None, region_masks=None,
                 region_prompt_embeds=None, region_image_embeds=None, output_type: str = "pil", return_dict: bool = True,
                 use_graph: bool = False, trajectory: Optional[list] = None, fusion_start: int = FUSION_START, dedup: bool = False,
                 cross_attention_kwargs=None, main_adapters=None, concept_adapters=None, lora_mode: str = "merged", eta: float = 0.0,
                 original_size=None, crops_coords_top_left=(0, 0), target_size=None, guess_mode: bool = False,
                 control_guidance_start=0.0, control_guidance_end=1.0, **kwargs):
        """``main_adapters`` / ``concept_adapters`` [(name, weight)]: LoRA adapters of ``concept_models.bank`` that PEFT would have active
        on the main / the concept pipe (inference_instantid.py:220-222 loads a style LoRA into both; nothing deactivates it).  The main
        UNet is called with ``cross_attention_kwargs`` (its ``"scale"`` is the LoRA scale, :596-616), the concept UNet with
        ``cross_attention_kwargs=None`` (:665-674): LoRA scale 1.0."""
        if prompt_embeds is None:
            raise L.OmgHipError("pass prompt_embeds=/pooled_prompt_embeds= (text encoders are outside this package's scope)")
        stage_cache = kwargs.pop("stage_cache", None)
        if eta != 0.0:
            raise L.OmgHipError("eta != 0 (stochastic DDIM) is not used by OMG and is not supported")
        if guess_mode:
            # the reference's OWN InstantID loop cannot run it: the IdentityNet is fed the two concept rows and its residuals are then concatenated with
            # zeros to FOUR rows for a two-row concept UNet (instantid_pipeline.py:638-657), and the t2i net's two-row residuals meet the four-row main
            # UNet without that concatenation (:580-616) — a shape error either way.  The LoRA flow's guess_mode is implemented (generate_many).
            raise L.OmgHipError("guess_mode=True: the reference's InstantID loop itself fails on it (residual batch 4 vs 2, instantid_pipeline.py:638-657); "
                                "LoraMultiConceptPipeline implements it")
        for dead in ("face_app", "prompt", "negative_prompt", "prompt_2", "negative_prompt_2", "clip_skip"):      # consumed by omg_amd.compat's front end
            kwargs.pop(dead, None)                                                                                 # (prompt encoding) when the scripts call it; dead here
        neg_cond = (kwargs.pop("negative_original_size", None), kwargs.pop("negative_crops_coords_top_left", (0, 0)), kwargs.pop("negative_target_size", None))
        callbacks = dict(callback=kwargs.pop("callback", None), callback_steps=kwargs.pop("callback_steps", None),
                         callback_on_step_end=kwargs.pop("callback_on_step_end", None),
                         callback_on_step_end_tensor_inputs=kwargs.pop("callback_on_step_end_tensor_inputs", ("latents",)))
        refuse_unimplemented({k: kwargs.pop(k) for k in list(kwargs) if k in _UNIMPLEMENTED}, kwargs, "InstantidMultiConceptPipeline.__call__")
        K = len(region_prompt_embeds or [])
        # stage 1 is called with image=None: no IdentityNet (instantid_pipeline.py:393, :426-428)
        use_idn = stage == 2 and image is not None
        req = dict(prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                   negative_pooled_prompt_embeds=negative_pooled_prompt_embeds, region_prompt_embeds=region_prompt_embeds or [],
                   region_masks=region_masks, latents=latents, generator=generator,
                   region_image_embeds=region_image_embeds if use_idn else None, kps_image=image if use_idn else None)
        traj_many = [] if trajectory is not None else None
        lat = self.generate_many([req], height=height, width=width, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                                 controller=controller, concept_models=concept_models or ConceptModels(self.unet, None), stage=stage,
                                 lora_list=[f"id{c}" for c in range(K)], styleL=False, use_graph=use_graph, trajectory=traj_many,
                                 fusion_start=fusion_start, identitynet=self.controlnet if use_idn else None,
                                 identitynet_conditioning_scale=controlnet_conditioning_scale,
                                 controlnet=self.controlnet2 if t2i_image is not None else None, controlnet_image=t2i_image,
                                 controlnet_conditioning_scale=t2i_controlnet_conditioning_scale, dedup=dedup, concept_lora=False,
                                 cross_attention_kwargs=cross_attention_kwargs, main_adapters=main_adapters,
                                 concept_adapters=concept_adapters, concept_adapter_scale=1.0, lora_mode=lora_mode,
                                 stage_cache=stage_cache, original_size=original_size, crops_coords_top_left=crops_coords_top_left,
                                 target_size=target_size, control_guidance_start=control_guidance_start, control_guidance_end=control_guidance_end,
                                 negative_original_size=neg_cond[0], negative_crops_coords_top_left=neg_cond[1], negative_target_size=neg_cond[2],
                                 **callbacks)[0]
        if trajectory is not None:
            trajectory.extend(t[0] for t in traj_many)
        lat = self._postprocess(lat, output_type)
        return StableDiffusionXLPipelineOutput(images=lat) if return_dict else (lat,)
