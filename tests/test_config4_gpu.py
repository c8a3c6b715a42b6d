"""BASELINE configs[4] as ONE workload (-m gpu): ControlNet on the main pass (lora_pipeline.py:519-536) + K = 3
concepts, one of them without a mask (:576-577 skips it), overlapping masks on the other two (:602 sums) + a style LoRA on the main
pass AND mixed into every concept pass at [0.7, 0.5] (inference_lora.py:162-164, lora_pipeline.py:588-591), 16-bit and in the
MX-fp8 arithmetic the config names — against the oracle's literal loop (oracle/pipeline.py with closures over oracle/unet.py and
oracle/controlnet.py).  Also the 50-step error-growth curve of the MX-fp8 mode next to round 2's fp16 / bf16 curves.

The UNet is the SDXL topology at widths (128, 256, 512): every transformer Linear and every resnet convolution of it is eligible
for the MX-fp8 path (K % 128 == 0), as in the full-size model."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from omg_amd import controller as pc
from omg_amd.controlnet import ControlNetModel
from omg_amd.lora import LoraAdapter, LoraBank
from omg_amd.pipeline import ConceptModels, LoraMultiConceptPipeline, revise_regionally_controlnet_forward
from omg_amd.schedulers import make_scheduler
from omg_amd.unet import UNet2DConditionModel, UNetConfig
from oracle import controller as oc
from oracle import controlnet as ocn
from oracle import pipeline as opipe
from oracle import schedulers as osched
from oracle import unet as ou

P = "a man and a woman walking on the street"
WIDE = dict(sample_size=16, block_out_channels=(128, 256, 512), transformer_layers_per_block=(1, 1, 2), attention_head_dim=(2, 4, 8),
            cross_attention_dim=128, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32)
dtype = torch.float16
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def emb(cfg, n, seed):
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(n, 77, cfg.cross_attention_dim, generator=g).to(dtype).float()
    p = torch.randn(n, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g).to(dtype).float()
    return e, p


def build(dev, with_controlnet=True):
    cfg, ocfg = UNetConfig(**WIDE), ou.UNetConfig(**WIDE)
    sd = ou.init_state_dict(ocfg, seed=0, dtype=dtype)
    unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev)
    unet.load_state_dict({k: v.to(dtype) for k, v in sd.items()})
    csd = cn = None
    if with_controlnet:
        csd = ocn.init_state_dict(ocfg, seed=3, dtype=dtype)
        cn = ControlNetModel(cfg, dtype=dtype, device=dev)
        cn.load_state_dict({k: v.to(dtype) for k, v in csd.items()})
    return cfg, ocfg, sd, unet, csd, cn


def set_precision(unet, mode, controlnet=None):
    """"mx8": every eligible layer class of the UNet — and, round 4, of the ControlNet of configs[4] (its blocks are the UNet's encoder
    blocks) — on the block-scaled fp8 MFMA; "fp16": all 16-bit."""
    from omg_amd.unet import MX8_CLASSES
    unet.set_linear_precision("mx8" if mode != "fp16" else "fp16")
    unet.set_conv_precision("mx8" if mode != "fp16" else "fp16")
    if controlnet is not None:
        n = controlnet.set_precision_classes(MX8_CLASSES if mode != "fp16" else ())
        assert (n > 0) == (mode != "fp16")


def test_config4_composition_matches_the_oracle_loop(dev):
    cfg, ocfg, sd, unet, csd, cn = build(dev)
    L = cfg.sample_size
    S, gs, fstart, cs = 8, 7.5, 3, 0.7
    H = W = L * 8
    pos_e, pos_p = emb(cfg, 1, 2); neg_e, neg_p = emb(cfg, 1, 1)
    pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    regions = []
    for c in range(3):
        re_, rp_ = emb(cfg, 2, 10 + c)
        regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
    m0 = torch.zeros(H, W); m0[H // 4:, W // 16: W // 2] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 16: W - 8] = 1          # overlaps m0 by 16 px: the sum rule
    masks = [m0, None, m2]                                                # concept 1 was not found by the segmenter
    pose = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(3)).to(dtype).float()
    lat0 = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(14))
    tid = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32)
    names = ou.lora_target_names(ocfg)
    ow, ofn = {}, {}
    for nm_, seed in (("c0", 100), ("c1", 101), ("c2", 102), ("style", 103)):
        ow[nm_], ofn[nm_] = ou.make_lora(ocfg, names, rank=8, seed=seed, scale=0.8, dtype=dtype)
    concept = ConceptModels(unet, LoraBank(unet, [LoraAdapter(k, {n: (a.to(dev), b.to(dev)) for n, (a, b) in w.items()}) for k, w in ow.items()]))
    args = ([P, P], S, {"default_": 1.0}, 0.4, L // 4, L // 4)
    pctl = pc.AttentionReplace(*args, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    req = dict(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp,
               region_prompt_embeds=regions, region_masks=masks, latents=lat0)

    def run(use_graph=False, dedup=False):
        pctl.reset()
        traj = []
        pipe.generate_many([req], height=H, width=W, num_inference_steps=S, guidance_scale=gs, cross_attention_kwargs={"scale": 0.8},
                           controller=pctl, concept_models=concept, stage=2, lora_list=["c0", "c1", "c2"], styleL=True, fusion_start=fstart,
                           controlnet=cn, controlnet_image=pose, controlnet_conditioning_scale=cs, trajectory=traj, use_graph=use_graph, dedup=dedup)
        assert (pctl.cur_step, pctl.cur_att_layer) == (S, 0)
        return torch.stack([t[0].cpu() for t in traj])

    # ---- the oracle's literal loop
    osch = osched.make("ddim", S)
    octl = oc.AttentionReplaceOracle(*args)
    octl.num_att_layers = pctl.num_att_layers
    attn_main = oc.reference_attn_fn(octl)
    ctx4, te4 = torch.cat([ne, pe]), torch.cat([npp, pp])

    def main(x, i):
        t = float(osch.timesteps[i])
        down, mid = ocn.controlnet_forward(csd, ocfg, x, t, ctx4, pose.repeat(4, 1, 1, 1), cs, te4, tid.repeat(4, 1))
        return ou.unet_forward(sd, ocfg, x, t, ctx4, te4, tid.repeat(4, 1), attn_fn=attn_main, lora=ofn["style"],
                               down_block_additional_residuals=down, mid_block_additional_residual=mid)

    def conc(c):
        ctx2 = torch.cat([regions[c][0], regions[c][1]]); te2 = torch.cat([regions[c][2], regions[c][3]])
        fn = lambda key, x: 0.7 * ofn[f"c{c}"](key, x) + 0.5 * ofn["style"](key, x)
        return lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx2, te2, tid.repeat(2, 1), lora=fn)

    rec = []
    ref = opipe.denoise(main, [conc(0), conc(1), conc(2)], osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=masks, fusion_start=fstart, record=rec)
    rec = torch.stack(rec)
    rms = ref.pow(2).mean().sqrt().item()

    # ---- 16-bit: parity, graph == eager, dedup == full
    set_precision(unet, "fp16")
    t16 = run()
    e16 = [(a - b).abs().max().item() / rms for a, b in zip(t16, rec)]
    print("configs[4] composition fp16: per-step max|d|/rms = " + " ".join(f"{e:.2e}" for e in e16))
    assert e16[-1] < 2e-2, e16
    assert torch.equal(run(use_graph=True), t16) and torch.equal(run(use_graph=True), t16)
    assert torch.equal(run(use_graph=True, dedup=True), t16)
    # every ingredient matters more than the tolerance: drop one at a time from the ORACLE and look at the change
    attn2 = oc.reference_attn_fn(_fresh(octl, args))
    no_cn = opipe.denoise(lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx4, te4, tid.repeat(4, 1), attn_fn=attn2, lora=ofn["style"]),
                          [conc(0), conc(1), conc(2)], osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=masks, fusion_start=fstart)
    assert (no_cn - ref).abs().max().item() / rms > 0.2
    # ---- MX-fp8 Linear + convolutions: same loop, measured tolerance (the oracle stays fp32)
    set_precision(unet, "mx8")
    try:
        t8_unet_only = run()
    finally:
        set_precision(unet, "fp16")
    set_precision(unet, "mx8", cn)                                  # round 4: the ControlNet of the main pass on the fp8 MFMA too
    try:
        t8 = run()
        assert torch.equal(run(use_graph=True), t8)                 # a precision switch must not replay the 16-bit graphs
        assert not torch.equal(t8, t16) and not torch.equal(t8, t8_unet_only)
    finally:
        set_precision(unet, "fp16", cn)
    assert torch.equal(run(use_graph=True), t16)                    # ... nor the other way round
    e8 = [(a - b).abs().max().item() / rms for a, b in zip(t8, rec)]
    r8 = [(a - b).pow(2).mean().sqrt().item() / rms for a, b in zip(t8, rec)]
    print("configs[4] composition MX-fp8: per-step max|d|/rms = " + " ".join(f"{e:.2e}" for e in e8))
    print("configs[4] composition MX-fp8: per-step rms(d)/rms  = " + " ".join(f"{e:.2e}" for e in r8))
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "r04_config4_loop_parity.json"), "w") as f:
            json.dump({"what": "8-step DDIM stage-2 call in BASELINE configs[4]'s composition (ControlNet on the main pass, 3 concepts with one None mask and "
                               "an overlap, style LoRA on main + concept passes), SDXL topology at widths (128, 256, 512), vs the fp32 oracle loop; per step, "
                               "relative to the rms of the oracle's final latents; mx8 = UNet AND ControlNet on the MX-fp8 MFMA (round 4), mx8_unet_only = round 3's mode",
                       "fp16_max": e16, "mx8_max": e8, "mx8_rms": r8,
                       "mx8_unet_only_max": [(a - b).abs().max().item() / rms for a, b in zip(t8_unet_only, rec)],
                       "mx8_unet_only_rms": [(a - b).pow(2).mean().sqrt().item() / rms for a, b in zip(t8_unet_only, rec)]}, f)
    except OSError:
        pass
    # measured on MI355X (profiles/r03_config4_loop_parity.json, UNet only): fp16 max 8.8e-3; MX-fp8 rms 0.13, max 0.57 of the latent rms at
    # step 8 (the error grows with the step count here because 8 steps are all inside the growth phase, cf. the 50-step curve); with the
    # ControlNet in fp8 too: profiles/r04_config4_loop_parity.json — bound = measured + ~50 %
    assert max(r8) < 0.25 and max(e8) < 1.0, (max(r8), max(e8))


def _fresh(octl, args):
    o = oc.AttentionReplaceOracle(*args)
    o.num_att_layers = octl.num_att_layers
    return o


@pytest.mark.slow
def test_fifty_step_error_growth_mx8(dev):
    """The 50-step curve VERDICT r2 asks for beside r02_error_growth_fp16.json: BASELINE configs[1]'s loop (50 DDIM steps, fusion for
    i > 15, 20-step self-replace window, guidance 7.5, two LoRA concepts with overlapping masks) with every eligible Linear and
    resnet convolution on the MX-fp8 MFMA, against the fp32 oracle loop; the 16-bit run of the same model beside it."""
    cfg, ocfg, sd, unet, _, _ = build(dev, with_controlnet=False)
    L = cfg.sample_size
    S, gs, fstart = 50, 7.5, 15
    H = W = L * 8
    pos_e, pos_p = emb(cfg, 1, 2); neg_e, neg_p = emb(cfg, 1, 1)
    pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    regions = []
    for c in range(2):
        re_, rp_ = emb(cfg, 2, 10 + c)
        regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
    m1 = torch.zeros(H, W); m1[H // 4:, W // 16: W // 2 - 8] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 24: W - 8] = 1
    masks = [m1, m2]
    lat0 = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(14))
    tid = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32)
    names = ou.lora_target_names(ocfg)
    ow, olora = [], []
    for c in range(2):
        w, fn = ou.make_lora(ocfg, names, rank=8, seed=100 + c, scale=0.8, dtype=dtype)
        ow.append(w); olora.append(fn)
    concept = ConceptModels(unet, LoraBank(unet, [LoraAdapter(f"c{c}", {k: (a.to(dev), b.to(dev)) for k, (a, b) in ow[c].items()}) for c in range(2)]))
    args = ([P, P], 50, {"default_": 1.0}, 0.4, L // 4, L // 4)
    pctl = pc.AttentionReplace(*args, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    osch = osched.make("ddim", S)
    octl = oc.AttentionReplaceOracle(*args)
    octl.num_att_layers = pctl.num_att_layers
    attn = oc.reference_attn_fn(octl)
    ctx4 = torch.cat([ne, pe]); te4 = torch.cat([npp, pp])

    def main(x, i):
        return ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx4, te4, tid.repeat(4, 1), attn_fn=attn)

    def conc(c):
        ctx2 = torch.cat([regions[c][0], regions[c][1]]); te2 = torch.cat([regions[c][2], regions[c][3]])
        return lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx2, te2, tid.repeat(2, 1), lora=olora[c])

    rec = []
    opipe.denoise(main, [conc(0), conc(1)], osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=masks, fusion_start=fstart, record=rec)

    def run():
        pctl.reset()
        traj = []
        pipe(output_type="latent", prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, height=H, width=W,
             num_inference_steps=S, guidance_scale=gs, latents=lat0, cross_attention_kwargs={"scale": 0.8}, controller=pctl,
             concept_models=concept, stage=2, region_masks=masks, lora_list=["c0", "c1"], styleL=False, region_prompt_embeds=regions,
             trajectory=traj, fusion_start=fstart)
        return [t.float().cpu() for t in traj]

    from omg_amd.unet import MX8_PRESETS
    curves = {}
    for mode in ("fp16", "mx8", "mx8-safe"):
        if mode == "mx8-safe":      # round 4: the three quietest layer classes of the sensitivity sweep (profiles/r04_mx8_sensitivity.json)
            assert unet.set_precision_classes(MX8_PRESETS["safe"]) > 0
        else:
            set_precision(unet, mode)
        try:
            traj = run()
        finally:
            set_precision(unet, "fp16")
        rms = [b.pow(2).mean().sqrt().item() for b in rec]
        curves[mode] = {"max_abs_over_rms": [(a - b).abs().max().item() / r for a, b, r in zip(traj, rec, rms)],
                        "rms_err_over_rms": [(a - b).pow(2).mean().sqrt().item() / r for a, b, r in zip(traj, rec, rms)]}
        pick = (0, 9, 15, 16, 19, 29, 39, 49)
        print(f"50-step stage-2 trajectory {mode}: max|d|/rms at steps 1,10,16,17,20,30,40,50 = " + " ".join(f"{curves[mode]['max_abs_over_rms'][i]:.2e}" for i in pick))
        print(f"50-step stage-2 trajectory {mode}: rms(d)/rms at the same steps             = " + " ".join(f"{curves[mode]['rms_err_over_rms'][i]:.2e}" for i in pick))
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "r04_error_growth_mx8.json"), "w") as f:
            json.dump({"what": "per-step error of a 50-step stage-2 call (SDXL topology at widths (128, 256, 512), DDIM, gs 7.5, fusion i>15, self-replace 20 "
                               "steps, 2 LoRA concepts with overlapping masks) vs the fp32 CPU oracle loop: fp16 storage, and MX-fp8 (OCP e4m3, E8M0 scale per "
                               "32) on every transformer Linear and resnet convolution (mx8) or on the three quietest layer classes only (mx8-safe: cross-attention "
                               "query / output projections and FF-out, omg_amd.unet.MX8_PRESETS)", **curves}, f)
    except OSError:
        pass
    assert max(curves["fp16"]["max_abs_over_rms"]) < 2e-2
    # measured on MI355X (profiles/r03_error_growth_mx8.json): fp16 5.3e-3 max / 1.5e-3 rms; MX-fp8 0.40 max / 0.104 rms of the latent rms,
    # both flat from step ~10 on (the fusion steps add nothing) — the loop-level tolerance of the fp8 mode, stated in bench.py's dtype string
    assert max(curves["mx8"]["rms_err_over_rms"]) < 0.16 and max(curves["mx8"]["max_abs_over_rms"]) < 0.65
    # "safe" preset: measured rms 3.1e-2 / max 0.11 (profiles/r04_mx8_sensitivity.json) — between the two, as the quadrature sum of its classes says
    assert max(curves["mx8-safe"]["rms_err_over_rms"]) < 0.05 and max(curves["mx8-safe"]["max_abs_over_rms"]) < 0.2
    assert max(curves["fp16"]["rms_err_over_rms"]) < max(curves["mx8-safe"]["rms_err_over_rms"]) < max(curves["mx8"]["rms_err_over_rms"])
