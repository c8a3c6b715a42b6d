"""B3 (-m gpu): the reference's own driver code — ``build_model_sd`` and ``sample_image`` of inference_lora.py (:152-171, :37-73),
transcribed verbatim (tests/test_compat_verbatim.py compares the syntax trees with /root/reference where it exists) under the script's
ORIGINAL import lines (``omg_amd.compat.install()`` registers ``src.pipelines.*``, ``src.prompt_attention.p2p_attention`` and
``diffusers``) — runs against omg_amd.compat on a synthetic model directory, stage 1 and stage 2,
with and without a style LoRA and a spatial condition; the result equals the lower-level embeddings-in / latents-out API fed with the
same encoders' outputs (so the string / PIL surface adds plumbing, not arithmetic)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import _fake_hub as hub

# ---- the reference's own import lines (inference_lora.py:29-32), UNCHANGED: omg_amd.compat.install() provides the modules ----------------
from omg_amd import compat as _compat
_compat.install()
from src.pipelines.lora_pipeline import LoraMultiConceptPipeline
from src.prompt_attention.p2p_attention import AttentionReplace
from diffusers import ControlNetModel, StableDiffusionXLPipeline
from src.pipelines.lora_pipeline import revise_regionally_controlnet_forward
_compat.uninstall()          # keep the aliases out of the other test modules of the process; the names above stay bound


# ---- inference_lora.py:37-73, unchanged ----------------------------------------------------------------------------------
def sample_image(pipe,
    input_prompt,
    input_neg_prompt=None,
    generator=None,
    concept_models=None,
    num_inference_steps=50,
    guidance_scale=7.5,
    controller=None,
    stage=None,
    region_masks=None,
    lora_list = None,
    styleL=None,
    **extra_kargs
):
    spatial_condition = extra_kargs.pop('spatial_condition')
    if spatial_condition is not None:
        spatial_condition_input = [spatial_condition] * len(input_prompt)
    else:
        spatial_condition_input = None

    images = pipe(
        prompt=input_prompt,
        concept_models=concept_models,
        negative_prompt=input_neg_prompt,
        generator=generator,
        guidance_scale=guidance_scale,
        num_inference_steps=num_inference_steps,
        cross_attention_kwargs={"scale": 0.8},
        controller=controller,
        stage=stage,
        region_masks=region_masks,
        lora_list=lora_list,
        styleL=styleL,
        image=spatial_condition_input,
        **extra_kargs).images

    return images


# ---- inference_lora.py:150-171, unchanged ---------------------------------------------------------------------------------
def build_model_sd(pretrained_model, controlnet_path, device, prompts, lora_paths, width, height, style_lora):
    controlnet = ControlNetModel.from_pretrained(controlnet_path, torch_dtype=torch.float16).to(device)
    pipe = LoraMultiConceptPipeline.from_pretrained(
        pretrained_model, controlnet=controlnet, torch_dtype=torch.float16, variant="fp16").to(device)
    controller = AttentionReplace(prompts, 50, cross_replace_steps={"default_": 1.}, self_replace_steps=0.4, tokenizer=pipe.tokenizer, device=device, dtype=torch.float16, width=width, height=height)
    revise_regionally_controlnet_forward(pipe.unet, controller)

    pipe_concept = StableDiffusionXLPipeline.from_pretrained(pretrained_model, torch_dtype=torch.float16, variant="fp16").to(device)
    pipe_concept.enable_xformers_memory_efficient_attention()

    if style_lora is not None and os.path.exists(style_lora):
        pipe.load_lora_weights(style_lora, weight_name="pytorch_lora_weights.safetensors", adapter_name='style')
        pipe_concept.load_lora_weights(style_lora, weight_name="pytorch_lora_weights.safetensors", adapter_name='style')

    pipe_list = []
    for lora_path in lora_paths.split('|'):
        adapter_name = lora_path.split('/')[-1].split('.')[0]
        pipe_concept.load_lora_weights(lora_path, weight_name="pytorch_lora_weights.safetensors", adapter_name=adapter_name)
        pipe_list.append(adapter_name)
    return pipe, controller, pipe_concept, pipe_list


@pytest.fixture(scope="module")
def hub_dirs(tmp_path_factory):
    from omg_amd import compat
    root = tmp_path_factory.mktemp("hub")
    model, cn = hub.write_sdxl_dir(str(root / "sdxl")), hub.write_controlnet_dir(str(root / "controlnet"))
    compat.clear_component_cache()
    comp = compat._components(model, torch.float16, "fp16")
    loras = [hub.write_lora_file(str(root / "loras" / f"{n}.safetensors"), comp.unet, 20 + i, text_encoders=[comp.text_encoder, comp.text_encoder_2])
             for i, n in enumerate(("chris-evans", "TaylorSwiftSDXL"))]
    style = os.path.dirname(hub.write_lora_file(str(root / "style" / "pytorch_lora_weights.safetensors"), comp.unet, 30, style="peft",
                                                text_encoders=[comp.text_encoder, comp.text_encoder_2]))
    compat.clear_component_cache()
    return model, cn, "|".join(loras), style


@pytest.mark.parametrize("use_style,use_cond", [(False, False), (True, False), (False, True)])
def test_reference_driver_code_runs_on_the_compat_objects(dev, hub_dirs, use_style, use_cond):
    from PIL import Image
    from omg_amd import compat
    model, cn, lora_paths, style = hub_dirs
    compat.clear_component_cache()
    device = dev
    prompt = "a man and a woman walking on the street"
    prompts = [prompt] * 2
    width = height = 128                                             # the tiny UNet's sample_size 16 x 8
    pipe, controller, pipe_concepts, pipe_list = build_model_sd(model, cn, device, prompts, lora_paths, width // 32, height // 32, style if use_style else None)
    styleL = use_style
    spatial = Image.fromarray((np.random.RandomState(0).rand(64, 64, 3) * 255).astype("uint8")) if use_cond else None
    kwargs = {'height': height, 'width': width, 'spatial_condition': spatial}
    region = [("a man in the park", "painting"), ("a woman in the park", "painting")]
    input_prompt = [prompts, region]
    S = 24                                                           # > 16 so that the i > 15 fusion branch fires (8 fused steps)
    image = sample_image(pipe, input_prompt=input_prompt, concept_models=pipe_concepts, input_neg_prompt=["painting"] * len(input_prompt),
                         generator=torch.Generator(device).manual_seed(7), controller=controller, stage=1, lora_list=pipe_list, styleL=styleL,
                         num_inference_steps=S, **kwargs)
    assert len(image) == 2 and image[0].size == (width, height) and image[0].mode == "RGB"
    assert np.array_equal(np.array(image[0]), np.array(image[1])), "stage 1: both samples are the same image"
    controller.reset()                                               # inference_lora.py:274
    assert pipe.tokenizer("man")["input_ids"][1] in pipe.tokenizer(prompt)["input_ids"][1:-1]
    mask1 = torch.zeros(height, width, dtype=torch.bool); mask1[32:, 8:60] = True           # what predict_mask returns: BoolTensor[H, W] | None
    mask2 = torch.zeros(height, width, dtype=torch.bool); mask2[32:, 56:120] = True
    image2 = sample_image(pipe, input_prompt=input_prompt, concept_models=pipe_concepts, input_neg_prompt=["painting"] * len(input_prompt),
                          generator=torch.Generator(device).manual_seed(7), controller=controller, stage=2, region_masks=[mask1, mask2],
                          lora_list=pipe_list, styleL=styleL, num_inference_steps=S, **kwargs)
    a0, b0, b1 = np.array(image[0]).astype(int), np.array(image2[0]).astype(int), np.array(image2[1]).astype(int)
    assert np.abs(a0 - b0).max() <= 1, "the base sample of stage 2 repeats stage 1 (same seed)"
    assert np.abs(b1 - b0).max() > 3, "the edited sample differs where the concepts were fused"

    # ---- the same call through the embeddings-in / latents-out API
    from omg_amd.pipeline import LoraMultiConceptPipeline as LowLevel
    te_scale = 0.8
    enc = pipe.encode_prompt
    pe, ne, pp, npp = enc(prompts, ["painting"] * 2, [("style", 1.0)] if styleL else None, te_scale)
    regs = []
    for name, (rp, rn) in zip(pipe_list, region):
        combo = [(name, 0.7), ("style", 0.5)] if styleL else [(name, 1.0)]
        rpe, rne, rpp, rnpp = enc(rp, rn, combo, te_scale)
        regs.append((rne, rpe, rnpp, rpp))
    low = LowLevel(pipe.unet, type(pipe.scheduler)(), vae_decode=pipe.vae.decode_latents)
    controller.reset()
    cond = None
    if use_cond:
        cond = torch.from_numpy(np.asarray(spatial.convert("RGB").resize((width, height))).astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    ref = low(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, region_prompt_embeds=regs,
              height=height, width=width, num_inference_steps=S, guidance_scale=7.5, generator=torch.Generator(device).manual_seed(7),
              cross_attention_kwargs={"scale": 0.8}, controller=controller, concept_models=pipe_concepts, stage=2, region_masks=[mask1, mask2],
              lora_list=pipe_list, styleL=styleL, image=cond, controlnet=pipe.controlnet if use_cond else None, output_type="pil").images
    assert np.array_equal(np.array(ref[1]), np.array(image2[1]))
