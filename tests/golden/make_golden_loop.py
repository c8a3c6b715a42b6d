"""Pin the loop / fusion / processor half of the oracle by EXECUTING THE REFERENCE'S OWN PIPELINE CODE in this container.

    python tests/golden/make_golden_loop.py       (needs /root/reference; writes loop_golden.npz next to itself)

``/root/reference/src/pipelines/lora_pipeline.py`` fails to import only because of its ``from diffusers ...`` / ``from torchvision ...``
lines (:16-57).  This script registers stand-in modules for exactly those names (test infrastructure — the same trick
``make_golden.py`` plays for ``cv2``), imports the file UNMODIFIED and drives the reference's own

  * ``LoraMultiConceptPipeline.__call__``            lora_pipeline.py:211-669  (prompt / batch layout, loop :485-632, fusion :568-607, CFG :610-612)
  * ``LoraMultiConceptPipeline.get_region_mask``     lora_pipeline.py:674-681
  * ``RegionControlNet_AttnProcessor.__call__``      lora_pipeline.py:61-133
  * ``revise_regionally_controlnet_forward``         lora_pipeline.py:136-152
  * ``AttentionReplace`` (the controller)            src/prompt_attention/p2p_attention.py (real, as in make_golden.py)

What the stand-ins supply — and therefore what this fixture does NOT pin (all third-party ``diffusers==0.25.0`` / ``peft==0.8.2`` code):
  * the base class ``StableDiffusionXLControlNetPipeline``: ``encode_prompt`` returns embeddings from a table keyed by the prompt string,
    ``prepare_latents`` multiplies the given latents by ``init_noise_sigma``, ``_get_add_time_ids`` concatenates its three tuples,
    ``progress_bar`` / ``maybe_free_model_hooks`` / ``upcast_vae`` do nothing;
  * the UNet: the ORACLE's functional forward (oracle/unet.py) for everything outside attention, behind an ``nn.Module`` tree whose attention
    layers are ``Attention``-protocol modules (class name 'Attention', ``to_q/to_k/to_v/to_out``, ``head_to_batch_dim``, ``get_attention_scores`` =
    softmax(baddbmm), ``set_processor``) so that the reference's installer walks it and the reference's processor + controller run for real at
    every attention layer of the main UNet; the concept UNet keeps a default processor = ``F.scaled_dot_product_attention`` (AttnProcessor2_0);
  * the scheduler: the oracle's DDIM / Euler behind the diffusers scheduler API (``set_timesteps``, ``timesteps``, ``scale_model_input``, ``step``);
  * ``concept_models``: ``set_adapters`` records the active (adapter, weight) list PEFT-style, the LoRA deltas are the oracle's synthetic ones.

The stored trajectories (per-step latents of stage 1 and stage 2) are compared in tests/test_oracle.py with ``oracle/pipeline.denoise`` +
``oracle/controller.reference_attn_fn`` to <= 1e-5: after that, A1 / A3 / A7 / A8 / A9 of the oracle are "pinned by the reference run here",
and every GPU loop test (which compares the HIP path with that oracle) inherits the pin.
"""
import contextlib
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import schedulers as osched  # noqa: E402
from oracle import unet as ou  # noqa: E402
from oracle.controller import WhitespaceTokenizer  # noqa: E402  (tokenizer only: no CLIP vocabulary offline)


# ----------------------------------------------------------------------------------------------------------------- stand-in modules
class _Cfg(dict):
    __getattr__ = dict.__getitem__


class StubPipelineBase:
    """What lora_pipeline.py uses of diffusers' StableDiffusionXLControlNetPipeline / DiffusionPipeline."""

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def register_to_config(self, **kw):
        self.config = _Cfg(kw)

    _execution_device = torch.device("cpu")
    guidance_scale = property(lambda s: s._guidance_scale)
    clip_skip = property(lambda s: s._clip_skip)
    cross_attention_kwargs = property(lambda s: s._cross_attention_kwargs)
    num_timesteps = property(lambda s: s._num_timesteps)

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1 and self.unet.config.time_cond_proj_dim is None

    embed_table = None       # prompt string -> (embeds (77, D), pooled (P,))

    def encode_prompt(self, prompt=None, prompt_2=None, device=None, num_images_per_prompt=1, do_classifier_free_guidance=True,
                      negative_prompt=None, negative_prompt_2=None, prompt_embeds=None, negative_prompt_embeds=None,
                      pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None, lora_scale=None, clip_skip=None):
        def look(p):
            ps = [p] if isinstance(p, str) else list(p)
            return torch.stack([self.embed_table[s][0] for s in ps]), torch.stack([self.embed_table[s][1] for s in ps])
        e, p = look(prompt)
        ne, np_ = look(negative_prompt)
        return e, ne, p, np_

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        assert latents is not None, "the fixture injects the initial noise (inference_lora.py:267 draws it on the device)"
        return latents.to(device) * self.scheduler.init_noise_sigma

    def prepare_extra_step_kwargs(self, generator, eta):
        return {}

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size, dtype, text_encoder_projection_dim=None):
        return torch.tensor([list(original_size + crops_coords_top_left + target_size)], dtype=dtype)

    def prepare_image(self, image, width, height, batch_size, num_images_per_prompt, device, dtype, do_classifier_free_guidance=False, guess_mode=False):
        """diffusers: control_image_processor.preprocess (PIL -> [0, 1] tensor; the fixture passes the tensor), repeat to the batch, x2 for CFG."""
        if isinstance(image, (list, tuple)):                       # inference_instantid.py:88-89 passes [image]
            assert len(image) == 1
            image = image[0]
        assert isinstance(image, torch.Tensor) and image.shape[0] == 1 and tuple(image.shape[-2:]) == (height, width)
        image = image.repeat_interleave(batch_size, dim=0).to(device=device, dtype=dtype)
        return torch.cat([image] * 2) if do_classifier_free_guidance and not guess_mode else image

    @contextlib.contextmanager
    def progress_bar(self, total=None):
        yield types.SimpleNamespace(update=lambda: None)

    def maybe_free_model_hooks(self):
        pass

    def check_inputs(self, *a, **k):          # instantid_pipeline.py:293 (lora_pipeline.py has the call commented out)
        pass

    def upcast_vae(self):
        pass


class StubControlNetModel(nn.Module):
    """diffusers ControlNetModel's call contract (lora_pipeline.py:519-536) over the oracle's functional ControlNet (oracle/controlnet.py);
    ``guess_mode`` = diffusers' logspace residual scaling (restated in oracle/controlnet.py, third-party arithmetic)."""
    config = _Cfg(global_pool_conditions=False)
    dtype = torch.float32

    def __init__(self, csd=None, cfg=None):
        super().__init__()
        self.csd, self.cfg, self.calls, self.scales_seen = csd, cfg, 0, []

    def forward(self, sample, timestep, encoder_hidden_states=None, controlnet_cond=None, conditioning_scale=1.0, guess_mode=False,
                added_cond_kwargs=None, return_dict=True):
        from oracle import controlnet as ocn
        assert return_dict is False
        self.calls += 1
        self.scales_seen.append(float(conditioning_scale))
        return ocn.controlnet_forward(self.csd, self.cfg, sample, float(timestep), encoder_hidden_states, controlnet_cond, conditioning_scale,
                                      added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"], guess_mode=guess_mode)


class StubMultiControlNetModel(nn.Module):
    """diffusers 0.25.0 ``MultiControlNetModel`` (third-party, recalled): ``.nets`` and a forward that calls every net with ITS image and ITS scale and
    sums the residuals in list order.  The reference wraps a list / tuple of ControlNets in it (lora_pipeline.py:175-176) and tests ``isinstance`` of it
    at :275-286, :366-383, :428."""
    dtype = torch.float32          # ModelMixin.dtype (read at lora_pipeline.py:377)

    def __init__(self, controlnets):
        super().__init__()
        self.nets = nn.ModuleList(controlnets)

    def forward(self, sample, timestep, encoder_hidden_states=None, controlnet_cond=None, conditioning_scale=None, guess_mode=False,
                added_cond_kwargs=None, return_dict=True):
        down = mid = None
        for i, (image, scale, net) in enumerate(zip(controlnet_cond, conditioning_scale, self.nets)):
            d, m = net(sample, timestep, encoder_hidden_states=encoder_hidden_states, controlnet_cond=image, conditioning_scale=scale,
                       guess_mode=guess_mode, added_cond_kwargs=added_cond_kwargs, return_dict=return_dict)
            if i == 0:
                down, mid = d, m
            else:
                down = [a + b for a, b in zip(down, d)]
                mid = mid + m
        return down, mid


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__golden_stub__ = True
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(sys.modules[parent], leaf, m)
        return m

    class _Any:                    # a name that is only ever imported / used as an annotation
        pass

    # transformers decides at ITS import whether torchvision exists: resolve the names lora_pipeline.py:8-14 imports before the stand-in is registered
    from transformers import CLIPImageProcessor, CLIPTextModel, CLIPTextModelWithProjection, CLIPTokenizer, CLIPVisionModelWithProjection  # noqa: F401
    log = types.SimpleNamespace(get_logger=lambda name: types.SimpleNamespace(warning=print, info=lambda *a, **k: None))
    mod("diffusers", StableDiffusionXLControlNetPipeline=StubPipelineBase)
    mod("diffusers.utils", USE_PEFT_BACKEND=True, deprecate=lambda *a, **k: None, logging=log, replace_example_docstring=lambda s: (lambda f: f),
        scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None, load_image=lambda *a, **k: None)
    mod("diffusers.utils.import_utils", is_invisible_watermark_available=lambda: False, is_xformers_available=lambda: False)
    mod("diffusers.utils.torch_utils", is_compiled_module=lambda m: False, is_torch_version=lambda op, v: True, randn_tensor=None)
    mod("diffusers.image_processor", PipelineImageInput=_Any, VaeImageProcessor=lambda **k: None)
    mod("diffusers.loaders", FromSingleFileMixin=_Any, IPAdapterMixin=_Any, StableDiffusionXLLoraLoaderMixin=_Any, TextualInversionLoaderMixin=_Any)
    mod("diffusers.models", AutoencoderKL=_Any, ControlNetModel=StubControlNetModel, ImageProjection=_Any, UNet2DConditionModel=_Any)
    mod("diffusers.models.attention_processor", AttnProcessor2_0=_Any, LoRAAttnProcessor2_0=_Any, LoRAXFormersAttnProcessor=_Any, XFormersAttnProcessor=_Any)
    mod("diffusers.models.lora", adjust_lora_scale_text_encoder=lambda *a, **k: None)
    mod("diffusers.schedulers", KarrasDiffusionSchedulers=_Any)
    mod("diffusers.pipelines")
    mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=StubPipelineBase)
    mod("diffusers.pipelines.stable_diffusion_xl", StableDiffusionXLPipelineOutput=lambda images: types.SimpleNamespace(images=images))
    mod("diffusers.pipelines.stable_diffusion_xl.pipeline_output", StableDiffusionXLPipelineOutput=lambda images: types.SimpleNamespace(images=images))
    mod("diffusers.pipelines.controlnet")
    mod("diffusers.pipelines.controlnet.multicontrolnet", MultiControlNetModel=StubMultiControlNetModel)
    mod("torchvision")
    mod("torchvision.transforms")
    mod("torchvision.transforms.functional", to_tensor=None)
    mod("torchvision.utils", save_image=None)
    # instantid_pipeline.py:757-768 get_face_embedding: load_image(path) -> cv2.cvtColor(np.array(img), COLOR_RGB2BGR) -> face_app.get(...)
    sys.modules["cv2"] = mod("cv2", cvtColor=lambda a, code: a[..., ::-1], COLOR_RGB2BGR=4)
    sys.modules["diffusers.utils"].load_image = lambda path: FACE_IMAGES[path]


FACE_IMAGES = {}         # "path" -> HxWx3 uint8 array standing for the reference photo


# --------------------------------------------------------------------------------------------- the UNet behind the diffusers module protocol
class LoraLinear(nn.Module):
    """PEFT-style wrapped Linear: base(x) + sum over the ACTIVE adapters of weight * scale * B(A(x)).  `state` is shared by the whole UNet and
    holds what set_adapters / the forward's cross_attention_kwargs["scale"] selected."""

    def __init__(self, weight, bias, key, state):
        super().__init__()
        self.weight, self.bias, self.key, self.state = weight, bias, key, state

    def forward(self, x):
        y = F.linear(x, self.weight, self.bias)
        return y + self.state.delta(self.key, x)


class LoraState:
    def __init__(self):
        self.adapters = {}           # name -> {key: (A, B)}
        self.active = ()             # ((name, weight), ...)
        self.scale = 1.0             # cross_attention_kwargs["scale"] of the running forward (PEFT: scale_lora_layers)

    def delta(self, key, x):
        d = 0.0
        for name, w in self.active:
            ab = self.adapters[name].get(key)
            if ab is not None:
                d = d + (w * self.scale) * F.linear(F.linear(x, ab[0]), ab[1])
        return d


class Attention(nn.Module):          # the class NAME is what revise_regionally_controlnet_forward looks for (lora_pipeline.py:139)
    """diffusers.models.attention_processor.Attention, the attributes and helpers the reference's processor reads (lora_pipeline.py:81-131)."""

    def __init__(self, sd, name, heads, state):
        super().__init__()
        lin = lambda n, b=None: LoraLinear(sd[f"{name}.{n}.weight"], sd.get(f"{name}.{n}.bias") if b else None, f"{name}.{n}", state)
        self.to_q, self.to_k, self.to_v = lin("to_q"), lin("to_k"), lin("to_v")
        self.to_out = nn.ModuleList([lin("to_out.0", True), nn.Dropout(0.0)])
        self.heads = heads
        self.scale = (self.to_q.weight.shape[0] // heads) ** -0.5
        self.group_norm = self.spatial_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = sdpa_processor

    def set_processor(self, p):
        self.processor = p

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        assert attention_mask is None
        return None

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        return t.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, n, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        return t.reshape(bh // self.heads, self.heads, n, d).permute(0, 2, 1, 3).reshape(bh // self.heads, n, d * self.heads)

    def get_attention_scores(self, query, key, attention_mask=None):
        assert attention_mask is None
        scores = torch.baddbmm(torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype), query, key.transpose(-1, -2),
                               beta=0, alpha=self.scale)
        return scores.softmax(dim=-1)

    def forward(self, hidden_states, encoder_hidden_states=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, **kw)


def sdpa_processor(attn, hidden_states, encoder_hidden_states=None, **kw):
    """AttnProcessor2_0 on (B, N, C) input: torch's own scaled_dot_product_attention."""
    src = hidden_states if encoder_hidden_states is None else encoder_hidden_states
    b = hidden_states.shape[0]
    split = lambda t: t.view(b, -1, attn.heads, t.shape[-1] // attn.heads).transpose(1, 2)
    o = F.scaled_dot_product_attention(split(attn.to_q(hidden_states)), split(attn.to_k(src)), split(attn.to_v(src)))
    o = o.transpose(1, 2).reshape(b, -1, attn.heads * o.shape[-1])
    return attn.to_out[1](attn.to_out[0](o))


class StubUNet(nn.Module):
    """down_blocks / mid_block / up_blocks hold the Attention modules under their diffusers names; forward = the oracle's functional UNet with
    every attention call routed through those modules (so through whatever processor the reference installed)."""

    def __init__(self, sd, cfg):
        super().__init__()
        self.sd, self.cfg, self.state = sd, cfg, LoraState()
        self.config = _Cfg(in_channels=cfg.in_channels, sample_size=cfg.sample_size, time_cond_proj_dim=None,
                           cross_attention_dim=cfg.cross_attention_dim, block_out_channels=cfg.block_out_channels)
        self.dtype, self.device = torch.float32, torch.device("cpu")
        self.by_name = {}
        heads = {}
        rev = list(reversed(cfg.attention_head_dim))
        for k in sd:
            if k.endswith(".to_q.weight"):
                name = k[: -len(".to_q.weight")]
                parts = name.split(".")
                h = cfg.attention_head_dim[int(parts[1])] if parts[0] == "down_blocks" else rev[int(parts[1])] if parts[0] == "up_blocks" else cfg.attention_head_dim[-1]
                heads[name] = h
        for name, h in heads.items():
            node = self
            parts = name.split(".")
            for p in parts[:-1]:
                if not hasattr(node, p):
                    node.add_module(p, nn.Module())
                node = getattr(node, p)
            a = Attention(sd, name, h, self.state)
            node.add_module(parts[-1], a)
            self.by_name[name] = a

    @property
    def attn_processors(self):          # diffusers: {"<module path>.processor": processor}; what instantid_single_pieline.py:186-213 walks
        return {name + ".processor": a.processor for name, a in self.by_name.items()}

    def set_attn_processor(self, procs):
        for name, a in self.by_name.items():
            a.set_processor(procs[name + ".processor"])

    def forward(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None, cross_attention_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, added_cond_kwargs=None, return_dict=True):
        assert timestep_cond is None and return_dict is False
        self.state.scale = (cross_attention_kwargs or {}).get("scale", 1.0)

        def attention(sd, name, heads, x, ctx, attn_fn, lora=None):
            return self.by_name[name](x, encoder_hidden_states=ctx)

        saved, ou.attention = ou.attention, attention
        try:
            out = ou.unet_forward(self.sd, self.cfg, sample, float(timestep), encoder_hidden_states, added_cond_kwargs["text_embeds"],
                                  added_cond_kwargs["time_ids"], lora=self.state.delta,
                                  down_block_additional_residuals=down_block_additional_residuals,
                                  mid_block_additional_residual=mid_block_additional_residual)
        finally:
            ou.attention = saved
        return (out,)


class SchedulerAdapter:
    """The oracle's scheduler behind the diffusers API (set_timesteps / timesteps / scale_model_input / step / order / init_noise_sigma)."""
    order = 1

    def __init__(self, kind, n_steps):
        self.o = osched.make(kind, n_steps)
        self.init_noise_sigma = self.o.init_noise_sigma
        self.n = n_steps

    def set_timesteps(self, n, device=None):
        assert n == self.n
        self.timesteps = torch.as_tensor(np.asarray(self.o.timesteps, dtype=np.float64))

    def _i(self, t):
        return int((self.timesteps == t).nonzero()[0, 0])

    def scale_model_input(self, x, t):
        return self.o.scale_model_input(x, self._i(t)).float()

    def step(self, noise, t, latents, return_dict=False, **kw):
        out = self.o.step(noise.double().numpy(), self._i(t), latents.double().numpy())
        return (torch.from_numpy(out).float(),)


class ConceptModels(StubPipelineBase):
    """The second pipeline object of inference_lora.py:166-170 as the loop uses it: unet, encode_prompt, _get_add_time_ids, set_adapters."""

    def __init__(self, unet, table):
        self.unet, self.embed_table = unet, table
        self.calls = []

    def set_adapters(self, names, adapter_weights=None):
        names = [names] if isinstance(names, str) else list(names)
        weights = adapter_weights or [1.0] * len(names)
        self.unet.state.active = tuple(zip(names, weights))
        self.calls.append(self.unet.state.active)


# ------------------------------------------------------------------------------------------------------------------- the cases
P = "a man and a woman walking on the street"
NEG = "blurry low quality"
REGION = [("a man wearing glasses", "ugly man"), ("a woman with red hair", "ugly woman"), ("a dog on a leash", "ugly dog")]


def embed_table(cfg, dtype=torch.float32):
    g = torch.Generator().manual_seed(77)
    t = {}
    for s in [P, NEG] + [x for r in REGION for x in r]:
        t[s] = (torch.randn(77, cfg.cross_attention_dim, generator=g).to(dtype).float(),
                torch.randn(cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g).to(dtype).float())
    return t


def masks_for(H, W, kind):
    m1 = torch.zeros(H, W); m1[H // 4:, W // 16: W // 2 - 8] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 24: W - 8] = 1         # overlaps m1 (the sum rule of :603)
    m3 = torch.zeros(H, W); m3[: H // 3, W // 3:] = 1
    return {"overlap": [m1, m2], "none_mid": [m1, None, m2], "three": [m1, m2, m3], "single": [m2], "all_none": [None, None]}[kind]


CASES = [  # name, scheduler, steps, guidance, mask kind, styleL, (latent h, latent w), flow
    ("ddim_overlap", "ddim", 20, 7.5, "overlap", False, (16, 16), "lora"),
    ("euler_none_mid", "euler", 20, 7.5, "none_mid", False, (16, 16), "lora"),
    ("ddim_style_three", "ddim", 18, 5.0, "three", True, (16, 16), "lora"),
    ("ddim_nonsquare", "ddim", 18, 7.5, "overlap", False, (24, 16), "lora"),
    ("ddim_single_concept", "ddim", 18, 7.5, "single", False, (16, 16), "lora"),
    ("euler_style_none_mid", "euler", 19, 3.0, "none_mid", True, (12, 20), "lora"),
    ("ddim_all_masks_none", "ddim", 18, 7.5, "all_none", False, (16, 16), "lora"),       # no concept pass runs: stage 2 == stage 1 (nothing is pasted, :581)
    # lora_pipeline.py:519-566 with `image` given: a ControlNet on the four main rows, none on the concept rows
    ("ddim_controlnet", "ddim", 18, 7.5, "overlap", False, (16, 16), "lora_cn"),
    # round 6 — the ControlNet kwargs the engine used to refuse (CN_VARIANTS below): control_guidance_start / _end -> controlnet_keep (:275-286, :421-428,
    # :511-517), guess_mode (:497-503, :531-535: the nets see the conditional rows only, zeros for the unconditional ones) and a LIST of ControlNets
    # (MultiControlNetModel, :175-176, :366-383: one image, scale and guidance window per net, residuals summed)
    ("ddim_controlnet_window", "ddim", 18, 7.5, "overlap", False, (16, 16), "lora_cn_window"),
    ("ddim_controlnet_guess", "ddim", 18, 7.5, "overlap", False, (16, 16), "lora_cn_guess"),
    ("euler_controlnet_multi", "euler", 18, 7.5, "none_mid", False, (16, 16), "lora_cn_multi"),
    # instantid_pipeline.py:477-483, :566-578: the ONE guidance window scales the IdentityNet (concept rows) AND the t2i ControlNet (main rows)
    ("euler_instantid_t2i_window", "euler", 20, 3.0, "overlap", False, (16, 16), "iid_t2i_window"),
    # instantid_pipeline.py:540-707: IdentityNet (key-point image + face tokens) and the IP-Adapter branch on the concept rows, guidance 3
    # (inference_instantid.py:78); `iid_t2i`: + a second ControlNet (self.controlnet2, t2i_image) on the main rows (:574-592)
    ("euler_instantid", "euler", 18, 3.0, "overlap", False, (16, 16), "iid"),
    ("euler_instantid_t2i", "euler", 18, 3.0, "none_mid", False, (16, 16), "iid_t2i"),
]
LORA_RANK, LORA_SEED0, LORA_SCALE = 8, 100, 0.8
CN_SCALE, IDN_SCALE, T2I_SCALE, IP_SCALE, IP_TOKENS, FACE_DIM = 0.7, 0.8, 0.6, 0.8, 16, 512
CN_VARIANTS = {"lora_cn": {}, "lora_cn_window": dict(control_guidance_start=0.2, control_guidance_end=0.7), "lora_cn_guess": dict(guess_mode=True),
               # two nets: the second one's own image, scale and window; both act in steps 6..9 of 18
               "lora_cn_multi": dict(control_guidance_start=[0.0, 0.3], control_guidance_end=[0.6, 1.0]),
               # 20 steps: both nets act in steps 4..17 — the first two fused steps (16, 17) with, the last two (18, 19) without the IdentityNet
               "iid_t2i_window": dict(control_guidance_start=0.2, control_guidance_end=0.9)}
RESAMPLER = dict(dim=1280, depth=4, dim_head=64, heads=20, num_queries=IP_TOKENS, embedding_dim=FACE_DIM)      # instantid_single_pieline.py:163-174


def resampler_state_dict(out_dim, seed=31):
    """Seeded weights of the InstantID image_proj_model (the reference's Resampler topology at full width; output = the tiny UNet's context width)."""
    from oracle import resampler as orsm
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in orsm.param_shapes(RESAMPLER["dim"], RESAMPLER["depth"], RESAMPLER["dim_head"], RESAMPLER["heads"], IP_TOKENS, FACE_DIM, out_dim).items():
        if k == "latents":
            sd[k] = torch.randn(shp, generator=g) * RESAMPLER["dim"] ** -0.5
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        elif "norm" in k or k.endswith(".1.0.weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) * shp[-1] ** -0.5
    return sd


def ip_weights(cfg, seed=9):
    """{attn2 module name: (to_k_ip, to_v_ip)} of the concept UNet's IPAttnProcessor2_0 layers."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, shp in ou.param_shapes(cfg).items():
        if name.endswith(".attn2.to_k.weight"):
            c_, cx = shp
            w[name[: -len(".to_k.weight")]] = (torch.randn(c_, cx, generator=g) * cx ** -0.5, torch.randn(c_, cx, generator=g) * cx ** -0.5)
    return w


def build(case):
    """Everything a case needs, shared by this script and tests/test_oracle.py (which re-creates the inputs and runs the ORACLE's loop)."""
    name, sched, steps, gs, mkind, style, (lh, lw), flow = case
    cfg = ou.UNetConfig.tiny()
    sd = ou.init_state_dict(cfg, seed=3)
    table = embed_table(cfg)
    H, W = lh * 8, lw * 8
    masks = masks_for(H, W, mkind)
    K = len(masks)
    names = ou.lora_target_names(cfg)
    loras = {f"c{c}": ou.make_lora(cfg, names, rank=LORA_RANK, seed=LORA_SEED0 + c, scale=1.0)[0] for c in range(K)}
    loras["style"] = ou.make_lora(cfg, names, rank=LORA_RANK, seed=LORA_SEED0 + 50, scale=1.0)[0]
    lat0 = torch.randn(1, 4, lh, lw, generator=torch.Generator().manual_seed(14))
    extra = {}
    if flow != "lora":
        from oracle import controlnet as ocn
        g = torch.Generator().manual_seed(3)
        extra = dict(csd=ocn.init_state_dict(cfg, seed=5), csd2=ocn.init_state_dict(cfg, seed=6), pose=torch.rand(1, 3, H, W, generator=g), pose2=torch.rand(1, 3, H, W, generator=g))
    if flow.startswith("iid"):
        g = torch.Generator().manual_seed(77)
        extra.update(rsd=resampler_state_dict(cfg.cross_attention_dim), ipw=ip_weights(cfg),
                     face_emb=[torch.randn(FACE_DIM, generator=g).numpy() for _ in range(K)])
    return dict(flow=flow, **extra, name=name, sched=sched, steps=steps, gs=gs, style=style, cfg=cfg, sd=sd, table=table, H=H, W=W, masks=masks, K=K, loras=loras, lat0=lat0,
                ctl_args=([P, P], 50 if name == "ddim_overlap" else steps, {"default_": 1.0}, 0.4, lw // 4, lh // 4))      # 50 = inference_lora.py:156


def run_reference(c):
    from src.pipelines.lora_pipeline import LoraMultiConceptPipeline, revise_regionally_controlnet_forward     # REFERENCE code
    from src.prompt_attention.p2p_attention import AttentionReplace                                          # REFERENCE code

    main_unet, concept_unet = StubUNet(c["sd"], c["cfg"]), StubUNet(c["sd"], c["cfg"])
    concept_unet.state.adapters = c["loras"]
    if c["style"]:                                         # inference_lora.py:162-164: the style LoRA is loaded into BOTH pipes
        main_unet.state.adapters = {"style": c["loras"]["style"]}
        main_unet.state.active = (("style", 1.0),)
    vae = nn.Module()
    vae.config = _Cfg(block_out_channels=(1, 2, 3, 4), force_upcast=True, scaling_factor=0.13025)
    vae.dtype = torch.float32
    vae.post_quant_conv = nn.Conv2d(4, 4, 1)
    cn = StubControlNetModel(c.get("csd"), c["cfg"])
    image = c["pose"] if c["flow"].startswith("lora_cn") else None
    cn_kw = dict(CN_VARIANTS.get(c["flow"], {}))
    cn_scale = CN_SCALE
    nets = [cn]
    if c["flow"] == "lora_cn_multi":             # the reference's constructor wraps the list in MultiControlNetModel (lora_pipeline.py:175-176)
        nets = [cn, StubControlNetModel(c["csd2"], c["cfg"])]
        cn, image, cn_scale = nets, [c["pose"], c["pose2"]], [CN_SCALE, T2I_SCALE]
    pipe = LoraMultiConceptPipeline(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None, unet=main_unet,
                                    controlnet=cn, scheduler=SchedulerAdapter(c["sched"], c["steps"]))
    pipe.embed_table = c["table"]
    concept = ConceptModels(concept_unet, c["table"])
    controller = AttentionReplace(*c["ctl_args"], tokenizer=WhitespaceTokenizer(), device="cpu", dtype=torch.float32)
    revise_regionally_controlnet_forward(pipe.unet, controller)
    out = {"num_att_layers": np.array(controller.num_att_layers)}
    for stage in (1, 2):
        controller.reset()
        traj = []
        real_step = pipe.scheduler.step

        def step(*a, **k):
            r = real_step(*a, **k)
            traj.append(r[0].clone())
            return r
        pipe.scheduler.step = step
        res = pipe(prompt=[[P, P], [REGION[k] for k in range(c["K"])]], negative_prompt=[NEG, NEG], image=image, height=c["H"], width=c["W"],
                   controlnet_conditioning_scale=cn_scale, **cn_kw,
                   num_inference_steps=c["steps"], guidance_scale=c["gs"], latents=c["lat0"].clone(), cross_attention_kwargs={"scale": LORA_SCALE},
                   controller=controller, concept_models=concept, stage=stage, region_masks=c["masks"], lora_list=[f"c{k}" for k in range(c["K"])],
                   styleL=c["style"], output_type="latent")
        pipe.scheduler.step = real_step
        assert torch.equal(res.images, traj[-1])
        assert (controller.cur_step, controller.cur_att_layer) == (c["steps"], 0)
        out[f"stage{stage}"] = torch.stack(traj).numpy()
    out["set_adapters_calls"] = np.array(len(concept.calls))
    out["controlnet_calls"] = np.array(nets[0].calls)
    if c["flow"] in ("lora_cn_window", "lora_cn_multi"):      # the conditioning_scale every call saw: scale * controlnet_keep[i], both stages back to back
        out["controlnet_scales_seen"] = np.array([n_.scales_seen for n_ in nets])
    return out


class FaceApp:
    """insightface's FaceAnalysis as get_face_embedding uses it (instantid_pipeline.py:757-768): .get(bgr image) -> [{bbox, embedding}, ...]."""

    def __init__(self, by_sum):
        self.by_sum = by_sum

    def get(self, bgr):
        emb = self.by_sum[int(bgr.astype(np.int64).sum())]
        # a second, SMALLER face comes first: the reference sorts by (x2 - x1) * y2 - y1 (sic, :764) and takes element [0] — "only use the maximum face"
        # in its comment, the smallest key in its code; the fixture's photo has one face per key value so that either reading picks the same one
        return [{"bbox": np.array([10.0, 10.0, 50.0, 60.0]), "embedding": emb}]


def run_reference_instantid(c):
    import tempfile
    from src.pipelines.instantid_pipeline import InstantidMultiConceptPipeline, revise_regionally_controlnet_forward     # REFERENCE code
    from src.pipelines.instantid_single_pieline import InstantidSingleConceptPipeline                                   # REFERENCE code
    from src.prompt_attention.p2p_attention import AttentionReplace                                                    # REFERENCE code

    main_unet, concept_unet = StubUNet(c["sd"], c["cfg"]), StubUNet(c["sd"], c["cfg"])
    vae = nn.Module()
    vae.config = _Cfg(block_out_channels=(1, 2, 3, 4), force_upcast=True, scaling_factor=0.13025)
    vae.dtype = torch.float32
    vae.post_quant_conv = nn.Conv2d(4, 4, 1)
    idn = StubControlNetModel(c["csd"], c["cfg"])
    pipe = InstantidMultiConceptPipeline(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None, unet=main_unet,
                                         controlnet=idn, scheduler=SchedulerAdapter(c["sched"], c["steps"]))
    pipe.embed_table = c["table"]
    pipe.controlnet2 = StubControlNetModel(c["csd2"], c["cfg"])                     # inference_instantid.py:198: pipe.controlnet2 = t2i ControlNet
    # the concept pipe is the reference's OWN single-concept class: its load_ip_adapter_instantid builds the reference's Resampler and installs
    # the reference's IPAttnProcessor2_0 / AttnProcessor2_0 (src/ip_adapter/attention_processor.py) on the concept UNet from a checkpoint file
    concept = InstantidSingleConceptPipeline()
    concept.unet, concept.embed_table = concept_unet, c["table"]
    concept.device, concept.dtype = torch.device("cpu"), torch.float32
    order = list(concept_unet.attn_processors)                                       # the checkpoint's "ip_adapter" keys are positional (ModuleList)
    ip_sd = {}
    for i, key in enumerate(order):
        mod = key[: -len(".processor")]
        if mod in c["ipw"]:
            ip_sd[f"{i}.to_k_ip.weight"], ip_sd[f"{i}.to_v_ip.weight"] = c["ipw"][mod]
    with tempfile.TemporaryDirectory() as td:
        ck = os.path.join(td, "ip-adapter.bin")
        torch.save({"image_proj": c["rsd"], "ip_adapter": ip_sd}, ck)
        concept.load_ip_adapter_instantid(ck, image_emb_dim=FACE_DIM, num_tokens=IP_TOKENS, scale=0.5)
    concept.set_ip_adapter_scale(IP_SCALE)                                           # inference_instantid.py:211-212
    photos = []
    for k in range(c["K"]):
        path = f"face{k}.png"
        FACE_IMAGES[path] = np.full((8, 8, 3), 40 + k, dtype=np.uint8)
        photos.append(path)
    face_app = FaceApp({int(FACE_IMAGES[pth].astype(np.int64).sum()): c["face_emb"][k] for k, pth in enumerate(photos)})
    controller = AttentionReplace(*c["ctl_args"], tokenizer=WhitespaceTokenizer(), device="cpu", dtype=torch.float32)
    revise_regionally_controlnet_forward(pipe.unet, controller)
    out = {"num_att_layers": np.array(controller.num_att_layers)}
    t2i = c["pose2"] if c["flow"].startswith("iid_t2i") else None
    cn_kw = dict(CN_VARIANTS.get(c["flow"], {}))
    for stage in (1, 2):
        controller.reset()
        traj = []
        real_step = pipe.scheduler.step

        def step(*a, **k):
            r = real_step(*a, **k)
            traj.append(r[0].clone())
            return r
        pipe.scheduler.step = step
        res = pipe(prompt=[[P, P], [REGION[k] + (photos[k],) for k in range(c["K"])]], negative_prompt=[NEG, NEG],
                   image=[c["pose"]] if stage == 2 else None, height=c["H"], width=c["W"], num_inference_steps=c["steps"], guidance_scale=c["gs"],
                   latents=c["lat0"].clone(), cross_attention_kwargs={"scale": LORA_SCALE}, controller=controller, concept_models=concept,
                   face_app=face_app, stage=stage, region_masks=c["masks"], controlnet_conditioning_scale=IDN_SCALE,
                   t2i_image=t2i, t2i_controlnet_conditioning_scale=T2I_SCALE, output_type="latent", **cn_kw)
        pipe.scheduler.step = real_step
        assert torch.equal(res.images, traj[-1])
        assert (controller.cur_step, controller.cur_att_layer) == (c["steps"], 0)
        out[f"stage{stage}"] = torch.stack(traj).numpy()
    out["set_adapters_calls"] = np.array(0)
    out["controlnet_calls"] = np.array(idn.calls)
    out["controlnet2_calls"] = np.array(pipe.controlnet2.calls)
    if cn_kw:      # the conditioning_scale every call of the two nets saw (IdentityNet: stage 2's fused steps, once per masked concept; t2i: every step of both stages)
        out["controlnet_scales_seen"] = np.array(idn.scales_seen)
        out["controlnet2_scales_seen"] = np.array(pipe.controlnet2.scales_seen)
    return out


def main():
    install_stubs()
    blob = {}
    for case in CASES:
        c = build(case)
        r = run_reference_instantid(c) if c["flow"].startswith("iid") else run_reference(c)
        for k, v in r.items():
            blob[f"{c['name']}/{k}"] = v
        d = np.abs(r["stage2"][-1][1] - r["stage1"][-1][1]).max()      # (0 when every mask is None)
        print(f"{c['name']}: {c['steps']} steps, layers {int(r['num_att_layers'])}, stage-2 edit vs stage-1 max|d| = {d:.3f}, "
              f"base sample equal: {np.abs(r['stage2'][-1][0] - r['stage1'][-1][0]).max():.2e}")
    np.savez_compressed(os.path.join(HERE, "loop_golden.npz"), **blob)
    print("wrote", os.path.join(HERE, "loop_golden.npz"), sum(v.nbytes for v in blob.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
