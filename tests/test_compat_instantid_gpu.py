"""B3 for the SECOND script BASELINE's north_star names (-m gpu): ``sample_image`` (:72-110), ``build_model_sd`` (:186-232),
``prepare_text`` (:234-255) and the two-stage main flow (:296-366) of the reference's inference_instantid.py, transcribed verbatim
(tests/test_compat_verbatim.py compares the syntax trees with /root/reference where it exists) under the script's ORIGINAL import lines
(``omg_amd.compat.install()``), run against omg_amd.compat on a synthetic model directory: stage 1 (``image=None`` => no IdentityNet,
instantid_pipeline.py:393, :426-428) and stage 2 (key-point image on every concept pass, face embeddings through
``concept_models._encode_prompt_image_emb``, :378-388), with and without ``pipe.controlnet2`` + ``t2i_image`` (:574-616).
The result equals the embeddings-in API (omg_amd.pipeline.InstantidMultiConceptPipeline) fed with the same encoders' outputs.

Not transcribed (third-party detectors between the stages): insightface's ``FaceAnalysis`` — a stand-in with the
same ``prepare`` / ``get`` surface returning bbox / kps / embedding — YOLO-World + EfficientViT-SAM (``predict_mask``: the masks
are given as the BoolTensor[H, W] it returns) and cv2's drawing calls inside ``draw_kps_multi`` (a PIL stand-in of the same
signature; the key-point image is an INPUT of the path)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import _fake_hub as hub

# ---- the reference's own import lines (inference_instantid.py:12, :34-37), UNCHANGED: omg_amd.compat.install() provides the modules ----
from omg_amd import compat as _compat
_compat.install()
from diffusers import ControlNetModel, StableDiffusionXLPipeline
from src.pipelines.instantid_pipeline import InstantidMultiConceptPipeline
from src.pipelines.instantid_single_pieline import InstantidSingleConceptPipeline
from src.prompt_attention.p2p_attention import AttentionReplace
from src.pipelines.instantid_pipeline import revise_regionally_controlnet_forward
_compat.uninstall()          # keep the aliases out of the other test modules of the process; the names above stay bound


class FaceAnalysis:
    """Stand-in for insightface.app.FaceAnalysis (inference_instantid.py:226-228): same constructor / prepare / get surface.  ``get``
    "detects" two faces whose embedding depends on the image content, so that different reference images give different identities."""
    calls = []

    def __init__(self, name=None, root=None, providers=None):
        self.name, self.root, self.providers = name, root, providers
        self.prepared = None

    def prepare(self, ctx_id=0, det_size=(640, 640)):
        self.prepared = (ctx_id, det_size)

    def get(self, bgr):
        assert self.prepared == (0, (640, 640)) and bgr.ndim == 3 and bgr.shape[2] == 3 and bgr.dtype == np.uint8
        FaceAnalysis.calls.append(bgr.shape)
        h, w = bgr.shape[:2]
        seed = int(bgr.astype(np.int64).sum() % (2 ** 31))
        rs = np.random.RandomState(seed)
        faces = []
        for k, (x0, x1) in enumerate(((0.1, 0.4), (0.55, 0.95))):
            cx, cy = (x0 + x1) / 2 * w, 0.3 * h
            kps = np.array([[cx - 8, cy - 6], [cx + 8, cy - 6], [cx, cy], [cx - 6, cy + 8], [cx + 6, cy + 8]], dtype=np.float32)
            faces.append({"bbox": np.array([x0 * w, 0.1 * h, x1 * w, 0.5 * h], dtype=np.float32), "kps": kps,
                          "embedding": rs.randn(512).astype(np.float32) * (1 + k)})
        return faces


def draw_kps_multi(image_pil, kps_list, color_list=[(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (255, 0, 255)]):
    """Same signature and role as inference_instantid.py:127-156 (an RGB canvas of the image's size with the 5 key points of every
    face); PIL instead of cv2, which is not installed here."""
    from PIL import Image, ImageDraw
    w, h = image_pil.size
    out = Image.new("RGB", (w, h), (0, 0, 0))
    d = ImageDraw.Draw(out)
    for kps in kps_list:
        for (x, y), c in zip(np.array(kps), color_list):
            d.ellipse([x - 3, y - 3, x + 3, y + 3], fill=c)
    return out


# ---- inference_instantid.py:72-110, unchanged -----------------------------------------------------------------------------
def sample_image(pipe,
    input_prompt,
    input_neg_prompt=None,
    generator=None,
    concept_models=None,
    num_inference_steps=50,
    guidance_scale=3,
    controller=None,
    face_app=None,
    image=None,
    stage=None,
    region_masks=None,
    controlnet_conditioning_scale=None,
    **extra_kargs
):

    if image is not None:
        image_condition = [image]
    else:
        image_condition = None


    images = pipe(
        prompt=input_prompt,
        concept_models=concept_models,
        negative_prompt=input_neg_prompt,
        generator=generator,
        guidance_scale=guidance_scale,
        num_inference_steps=num_inference_steps,
        cross_attention_kwargs={"scale": 0.8},
        controller=controller,
        image=image_condition,
        face_app=face_app,
        stage=stage,
        controlnet_conditioning_scale=controlnet_conditioning_scale,
        region_masks=region_masks,
        **extra_kargs).images
    return images


# ---- inference_instantid.py:186-232, unchanged ------------------------------------------------------------------------------
def build_model_sd(pretrained_model, controlnet_path, face_adapter, device, prompts, antelopev2_path, width, height, style_lora, condition_checkpoint, adapter_ratio):
    controlnet = ControlNetModel.from_pretrained(controlnet_path, torch_dtype=torch.float16)
    pipe = InstantidMultiConceptPipeline.from_pretrained(
        pretrained_model, controlnet=controlnet, torch_dtype=torch.float16, variant="fp16").to(device)

    controller = AttentionReplace(prompts, 50, cross_replace_steps={"default_": 1.},
                                  self_replace_steps=0.4, tokenizer=pipe.tokenizer, device=device, width=width, height=height,
                                  dtype=torch.float16)
    revise_regionally_controlnet_forward(pipe.unet, controller)

    controlnet_concept = ControlNetModel.from_pretrained(controlnet_path, torch_dtype=torch.float16)
    pipe_concept = InstantidSingleConceptPipeline.from_pretrained(
        pretrained_model,
        controlnet=controlnet_concept,
        torch_dtype=torch.float16
    )
    pipe_concept.load_ip_adapter_instantid(face_adapter)
    pipe_concept.set_ip_adapter_scale(adapter_ratio)
    pipe_concept.to(device)
    pipe_concept.image_proj_model.to(pipe_concept._execution_device)

    if condition_checkpoint is not None and os.path.exists(condition_checkpoint):
        t2i_controlnet = ControlNetModel.from_pretrained(condition_checkpoint, torch_dtype=torch.float16).to(device)
        pipe.controlnet2 = t2i_controlnet

    if style_lora is not None and os.path.exists(style_lora):
        pipe.load_lora_weights(style_lora, weight_name="pytorch_lora_weights.safetensors", adapter_name='style')
        pipe_concept.load_lora_weights(style_lora, weight_name="pytorch_lora_weights.safetensors", adapter_name='style')


    # modify
    app = FaceAnalysis(name='antelopev2', root=antelopev2_path,
                       providers=['CUDAExecutionProvider', 'CPUExecutionProvider'])
    app.prepare(ctx_id=0, det_size=(640, 640))

    return pipe, controller, pipe_concept, app


# ---- inference_instantid.py:234-255, unchanged ------------------------------------------------------------------------------
def prepare_text(prompt, region_prompts):
    region_collection = []

    regions = region_prompts.split('|')

    for region in regions:
        if region == '':
            break
        prompt_region, neg_prompt_region, ref_img = region.split('-*-')
        prompt_region = prompt_region.replace('[', '').replace(']', '')
        neg_prompt_region = neg_prompt_region.replace('[', '').replace(']', '')

        region_collection.append((prompt_region, neg_prompt_region, ref_img))
    return (prompt, region_collection)


def diffusers_attn_processor_names(ucfg):
    """The order of diffusers' ``unet.attn_processors`` written out from the block structure, independently of the package's helper:
    down blocks, then up blocks, then the mid block; attn1 before attn2 inside every transformer block."""
    tl = ucfg["transformer_layers_per_block"]
    names = []
    for b, kind in enumerate(ucfg["down_block_types"]):
        if kind.startswith("CrossAttn"):
            for a in range(ucfg["layers_per_block"]):
                for t in range(tl[b]):
                    names += [f"down_blocks.{b}.attentions.{a}.transformer_blocks.{t}.attn{k}" for k in (1, 2)]
    rtl = list(reversed(tl))
    for b, kind in enumerate(ucfg["up_block_types"]):
        if kind.startswith("CrossAttn"):
            for a in range(ucfg["layers_per_block"] + 1):
                for t in range(rtl[b]):
                    names += [f"up_blocks.{b}.attentions.{a}.transformer_blocks.{t}.attn{k}" for k in (1, 2)]
    for t in range(tl[-1]):
        names += [f"mid_block.attentions.0.transformer_blocks.{t}.attn{k}" for k in (1, 2)]
    return names


@pytest.fixture(scope="module")
def hub_dirs(tmp_path_factory):
    """Model directory (both weight-file variants, as a real SDXL directory has), IdentityNet and pose-ControlNet directories, an
    ``ip-adapter.bin`` = {"image_proj": Resampler state dict, "ip_adapter": {"<i>.to_k_ip.weight", "<i>.to_v_ip.weight"}} with i the
    position in diffusers' attn_processors order (instantid_single_pieline.py:179-182, :208-212), and two reference face images."""
    from PIL import Image
    from omg_amd import compat
    from omg_amd.resampler import Resampler
    root = tmp_path_factory.mktemp("hub_iid")
    model = hub.write_sdxl_dir(str(root / "sdxl"), plain_copies=True)
    idn = hub.write_controlnet_dir(str(root / "identitynet"), seed=7)
    pose = hub.write_controlnet_dir(str(root / "pose"), seed=8)
    cx = hub.UNET_CFG["cross_attention_dim"]
    g = torch.Generator().manual_seed(11)
    res = Resampler(dim=1280, depth=4, dim_head=64, heads=20, num_queries=16, embedding_dim=512, output_dim=cx, ff_mult=4, dtype=torch.float16, device="cpu")
    image_proj = hub._rand_state(res, 12)
    ip = {}
    widths = {"down_blocks.1": 128, "down_blocks.2": 256, "up_blocks.0": 256, "up_blocks.1": 128, "mid_block": 256}
    for i, name in enumerate(diffusers_attn_processor_names(hub.UNET_CFG)):
        if name.endswith("attn2"):
            c = widths[".".join(name.split(".")[:2]) if not name.startswith("mid") else "mid_block"]
            ip[f"{i}.to_k_ip.weight"] = (torch.randn(c, cx, generator=g) * cx ** -0.5).half()
            ip[f"{i}.to_v_ip.weight"] = (torch.randn(c, cx, generator=g) * cx ** -0.5).half()
    ckpt = str(root / "ip-adapter.bin")
    torch.save({"image_proj": image_proj, "ip_adapter": ip}, ckpt)
    refs = []
    for k in range(2):
        p = str(root / f"ref{k}.jpg")
        Image.fromarray((np.random.RandomState(40 + k).rand(96, 80, 3) * 255).astype("uint8")).save(p)
        refs.append(p)
    compat.clear_component_cache()
    return model, idn, pose, ckpt, refs, str(root / "antelopev2")


@pytest.mark.parametrize("use_pose", [False, True])
def test_instantid_driver_code_runs_on_the_compat_objects(dev, hub_dirs, use_pose):
    from PIL import Image
    from omg_amd import compat
    model, idn_dir, pose_dir, ckpt, refs, antelope = hub_dirs
    compat.clear_component_cache()
    FaceAnalysis.calls.clear()
    device = dev
    prompt = "a man and a woman walking on the street"
    prompts = [prompt] * 2
    width = height = 128
    spatial_condition = Image.fromarray((np.random.RandomState(5).rand(64, 64, 3) * 255).astype("uint8")).resize((width, height)) if use_pose else None
    kwargs = {'height': height, 'width': width, 't2i_image': spatial_condition, 't2i_controlnet_conditioning_scale': 0.7}
    pipe, controller, pipe_concepts, face_app = build_model_sd(model, idn_dir, ckpt, device, list(prompts), antelope, width // 32, height // 32, None,
                                                               pose_dir if use_pose else None, 0.8)
    assert pipe_concepts._unet is pipe.unet, "main and concept pipe share ONE UNet although the script loads them with different variants"
    assert (getattr(pipe, "controlnet2", None) is not None) == use_pose and face_app.prepared is not None
    rewrite = f"[a man in the park]-*-[painting]-*-{refs[0]}|[a woman in the park]-*-[painting]-*-{refs[1]}"
    input_prompt = [prepare_text(p, p_w) for p, p_w in zip(prompts, [rewrite])]
    input_prompt = [prompts, input_prompt[0][1]]
    assert input_prompt[1][0] == ("a man in the park", "painting", refs[0])
    S, cfg_scale, idn_rate, seed = 24, 3.0, 0.8, 7                     # > 16 steps so that the i > 15 fusion branch fires

    image = sample_image(pipe, input_prompt=input_prompt, concept_models=pipe_concepts, input_neg_prompt=["painting"] * len(input_prompt),
                         generator=torch.Generator(device).manual_seed(seed), controller=controller, face_app=face_app,
                         controlnet_conditioning_scale=idn_rate, stage=1, guidance_scale=cfg_scale, num_inference_steps=S, **kwargs)
    assert len(image) == 2 and image[0].size == (width, height) and image[0].mode == "RGB"
    assert np.array_equal(np.array(image[0]), np.array(image[1])), "stage 1: both samples are the same image"
    assert FaceAnalysis.calls == [], "stage 1 looks at no face (instantid_pipeline.py:376: embeddings only when stage == 2)"
    controller.reset()
    assert pipe.tokenizer("man")["input_ids"][1] in pipe.tokenizer(prompt)["input_ids"][1:-1]
    assert pipe.tokenizer("woman")["input_ids"][1] in pipe.tokenizer(prompt)["input_ids"][1:-1]
    mask1 = torch.zeros(height, width, dtype=torch.bool); mask1[32:, 8:60] = True           # what predict_mask returns: BoolTensor[H, W] | None
    mask2 = torch.zeros(height, width, dtype=torch.bool); mask2[32:, 56:120] = True
    # :353-354 (cv2.cvtColor(np.array(image[0]), cv2.COLOR_RGB2BGR) written as the channel flip it is)
    face_info = face_app.get(np.ascontiguousarray(np.array(image[0])[:, :, ::-1]))
    face_kps = draw_kps_multi(image[0], [face['kps'] for face in face_info])
    image2 = sample_image(pipe, input_prompt=input_prompt, concept_models=pipe_concepts, input_neg_prompt=["painting"] * len(input_prompt),
                          generator=torch.Generator(device).manual_seed(seed), controller=controller, face_app=face_app, image=face_kps, stage=2,
                          controlnet_conditioning_scale=idn_rate, region_masks=[mask1, mask2], guidance_scale=cfg_scale, num_inference_steps=S, **kwargs)
    assert len(FaceAnalysis.calls) == 3 and FaceAnalysis.calls[1:] == [(96, 80, 3)] * 2, "one detection per reference image inside the stage-2 call"
    a0, b0, b1 = np.array(image[0]).astype(int), np.array(image2[0]).astype(int), np.array(image2[1]).astype(int)
    assert np.abs(a0 - b0).max() <= 1, "the base sample of stage 2 repeats stage 1 (same seed)"
    assert np.abs(b1 - b0).max() > 3, "the edited sample differs where the identities were fused"

    # ---- the same stage-2 call through the embeddings-in API
    from omg_amd.pipeline import InstantidMultiConceptPipeline as LowLevel
    regions = input_prompt[1]
    enc = pipe.encode_prompt
    pe, ne, pp, npp = enc(list(prompts) + [r[0] for r in regions], ["painting"] * 2 + [r[1] for r in regions], None, 0.8)
    embs = compat.get_face_embedding(face_app, [r[2] for r in regions])
    assert not np.allclose(embs[0], embs[1])
    tokens = [pipe_concepts._encode_prompt_image_emb(e, pipe_concepts._execution_device, 1, pipe.unet.dtype, True) for e in embs]
    assert tuple(tokens[0].shape) == (2, 16, hub.UNET_CFG["cross_attention_dim"])
    low = LowLevel(pipe.unet, pipe.controlnet, type(pipe.scheduler)(), controlnet2=getattr(pipe, "controlnet2", None), vae_decode=pipe.vae.decode_latents)
    controller.reset()
    to_t = lambda im: torch.from_numpy(np.asarray(im.convert("RGB").resize((width, height))).astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    ref = low(prompt_embeds=pe[:2], negative_prompt_embeds=ne[:2], pooled_prompt_embeds=pp[:2], negative_pooled_prompt_embeds=npp[:2],
              region_prompt_embeds=[(ne[2 + c: 3 + c], pe[2 + c: 3 + c], npp[2 + c: 3 + c], pp[2 + c: 3 + c]) for c in range(2)],
              region_image_embeds=tokens, image=to_t(face_kps), t2i_image=to_t(spatial_condition) if use_pose else None,
              height=height, width=width, num_inference_steps=S, guidance_scale=cfg_scale, generator=torch.Generator(device).manual_seed(seed),
              controlnet_conditioning_scale=idn_rate, t2i_controlnet_conditioning_scale=0.7, controller=controller, concept_models=pipe_concepts,
              stage=2, region_masks=[mask1, mask2], output_type="pil").images
    assert np.array_equal(np.array(ref[1]), np.array(image2[1]))
    # the identities matter: swapping the two reference images changes the edited sample
    controller.reset()
    swapped = [prompts, [(regions[0][0], regions[0][1], refs[1]), (regions[1][0], regions[1][1], refs[0])]]
    image3 = sample_image(pipe, input_prompt=swapped, concept_models=pipe_concepts, input_neg_prompt=["painting"] * len(input_prompt),
                          generator=torch.Generator(device).manual_seed(seed), controller=controller, face_app=face_app, image=face_kps, stage=2,
                          controlnet_conditioning_scale=idn_rate, region_masks=[mask1, mask2], guidance_scale=cfg_scale, num_inference_steps=S, **kwargs)
    assert not np.array_equal(np.array(image3[1]), np.array(image2[1]))


def test_a_style_lora_loaded_by_the_script_is_active_on_both_pipes(dev, hub_dirs, tmp_path):
    """inference_instantid.py:220-222 loads the style LoRA into BOTH pipes and nothing in instantid_pipeline.py ever calls
    ``set_adapters``: PEFT leaves the freshly loaded adapter switched on, so the reference's images carry it — main rows at the caller's
    ``cross_attention_kwargs["scale"]`` (0.8 in ``sample_image``, :596-616), concept rows at 1.0 (``cross_attention_kwargs=None``, :665-674),
    text encoders at ``lora_scale`` = 0.8 (:330-360).  Round 3 loaded the file and ignored it."""
    from PIL import Image
    from omg_amd import compat
    model, idn_dir, pose_dir, ckpt, refs, antelope = hub_dirs
    compat.clear_component_cache()
    comp = compat._components(model, torch.float16, None)
    style = os.path.dirname(hub.write_lora_file(str(tmp_path / "style" / "pytorch_lora_weights.safetensors"), comp.unet, 31, style="peft",
                                                text_encoders=[comp.text_encoder, comp.text_encoder_2]))
    device = dev
    prompt = "a man and a woman walking on the street"
    prompts = [prompt] * 2
    width = height = 128
    S, cfg_scale, idn_rate, seed = 20, 3.0, 0.8, 7
    rewrite = f"[a man in the park]-*-[painting]-*-{refs[0]}|[a woman in the park]-*-[painting]-*-{refs[1]}"
    input_prompt = [prompts, prepare_text(prompt, rewrite)[1]]
    mask1 = torch.zeros(height, width, dtype=torch.bool); mask1[32:, 8:60] = True
    mask2 = torch.zeros(height, width, dtype=torch.bool); mask2[32:, 56:120] = True
    kps = Image.fromarray((np.random.RandomState(6).rand(height, width, 3) * 255).astype("uint8"))
    kwargs = {'height': height, 'width': width, 't2i_image': None, 't2i_controlnet_conditioning_scale': 0.7}
    out = {}
    for tag, style_dir in (("plain", None), ("style", style)):
        compat.clear_component_cache()
        pipe, controller, pipe_concepts, face_app = build_model_sd(model, idn_dir, ckpt, device, list(prompts), antelope, width // 32, height // 32,
                                                                   style_dir, None, 0.8)
        assert pipe.peft_active_adapters() == ([("style", 1.0)] if style_dir else []) == pipe_concepts.peft_active_adapters()
        out[tag] = sample_image(pipe, input_prompt=input_prompt, concept_models=pipe_concepts, input_neg_prompt=["painting"] * 2,
                                generator=torch.Generator(device).manual_seed(seed), controller=controller, face_app=face_app, image=kps, stage=2,
                                controlnet_conditioning_scale=idn_rate, region_masks=[mask1, mask2], guidance_scale=cfg_scale,
                                num_inference_steps=S, **kwargs)
    a, b = np.array(out["plain"][1]).astype(int), np.array(out["style"][1]).astype(int)
    assert np.abs(a - b).max() > 3, "the style adapter changes the image (it was silently ignored in round 3)"
    assert np.abs(np.array(out["plain"][0]).astype(int) - np.array(out["style"][0]).astype(int)).max() > 3, "... on the base sample too (main rows)"

    # ---- the same call through the embeddings-in API with the adapters named explicitly (pipe / pipe_concepts are the style build)
    from omg_amd.pipeline import InstantidMultiConceptPipeline as LowLevel
    regions = input_prompt[1]
    pe, ne, pp, npp = pipe.encode_prompt(list(prompts) + [r[0] for r in regions], ["painting"] * 2 + [r[1] for r in regions], [("style", 1.0)], 0.8)
    pe0 = pipe.encode_prompt(list(prompts), ["painting"] * 2, None, 0.8)[0]
    assert not torch.equal(pe0, pe[:2]), "the adapter's text-encoder half acts while the prompts are encoded"
    tokens = [pipe_concepts._encode_prompt_image_emb(e, pipe_concepts._execution_device, 1, pipe.unet.dtype, True)
              for e in compat.get_face_embedding(face_app, [r[2] for r in regions])]
    low = LowLevel(pipe.unet, pipe.controlnet, type(pipe.scheduler)(), vae_decode=pipe.vae.decode_latents)
    controller.reset()
    to_t = lambda im: torch.from_numpy(np.asarray(im.convert("RGB").resize((width, height))).astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    ref = low(prompt_embeds=pe[:2], negative_prompt_embeds=ne[:2], pooled_prompt_embeds=pp[:2], negative_pooled_prompt_embeds=npp[:2],
              region_prompt_embeds=[(ne[2 + c: 3 + c], pe[2 + c: 3 + c], npp[2 + c: 3 + c], pp[2 + c: 3 + c]) for c in range(2)],
              region_image_embeds=tokens, image=to_t(kps), height=height, width=width, num_inference_steps=S, guidance_scale=cfg_scale,
              generator=torch.Generator(device).manual_seed(seed), controlnet_conditioning_scale=idn_rate, controller=controller,
              concept_models=pipe_concepts, stage=2, region_masks=[mask1, mask2], output_type="pil",
              cross_attention_kwargs={"scale": 0.8}, main_adapters=[("style", 1.0)], concept_adapters=[("style", 1.0)]).images
    assert np.array_equal(np.array(ref[1]), np.array(out["style"][1]))
