"""Measurements beside the headline bench line (run on the GPU box, results under profiles/):

  ips      images/s of the headline workload at 1, 4 and 8 lock-step requests per step (1 = the reference's own call shape)
  both     stage 1 + stage 2 per image: 50 plain steps, then the stage-2 call; segmentation excluded
  instantid  BASELINE configs[2]: OMG + InstantID, 2 identities, 1024^2, 30 Euler steps, guidance 3.0: IdentityNet (ControlNet)
             on every concept pass + IP-Adapter cross-attention with 16 face tokens

  config4  BASELINE configs[4]: OMG + ControlNet (openpose-sdxl architecture, on the main pass of every step) + 3 concepts + style LoRA
           (main pass, and [0.7, 0.5] with each concept), 1024^2, 50 DDIM steps; --dtype fp8 = the config's arithmetic (MX-fp8 on the
           transformer Linears and resnet convolutions of the UNet and, round 4, of the ControlNet): 3.337 PFLOP per image

python tools/bench_extras.py ips|both|instantid|config4 [--steps K] [--dtype fp16|fp8]   -> one JSON line per measurement
"""
import argparse, contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import controller as pc
from omg_amd.pipeline import ConceptModels, LoraMultiConceptPipeline, revise_regionally_controlnet_forward
from omg_amd.schedulers import make_scheduler
from omg_amd.synthetic import c2_inputs, c2_masks, make_concept_models
from omg_amd.unet import UNet2DConditionModel, UNetConfig

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["ips", "both", "instantid", "config4"])
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp8"])
ap.add_argument("--ips", default="1,4,8")
a = ap.parse_args()
dev, dt = torch.device("cuda:0"), torch.float16
unet = UNet2DConditionModel(UNetConfig.sdxl(), dtype=dt, device=dev).init_synthetic_(seed=0)
if a.dtype == "fp8":
    unet.set_linear_precision("mx8")
    if a.what == "config4":
        unet.set_conv_precision("mx8")


def random_init_(module, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in module.named_parameters():
        if name.endswith(".weight") and p.dim() >= 2:
            w = torch.randn(p.shape, generator=g, device=dev) * p[0].numel() ** -0.5
        elif name.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g, device=dev)
        else:
            w = 0.1 * torch.randn(p.shape, generator=g, device=dev)
        p.data.copy_(w.to(p.dtype))
    module.invalidate_packed()
    return module

P = "a man and a woman walking on the street"
ctl = pc.AttentionReplace([P, P], 50, {"default_": 1.0}, 0.4, 32, 32, device=dev, dtype=dt)
with contextlib.redirect_stdout(io.StringIO()):
    revise_regionally_controlnet_forward(unet, ctl)
masks = c2_masks(1024, 1024, device=dev)
from omg_amd.vae import AutoencoderKLDecoder, VaeConfig
vae = AutoencoderKLDecoder(VaeConfig.sdxl(), dtype=torch.float16, device=dev, upcast=True).init_synthetic_(seed=1)      # the reference's upcast decode, as bench.py


def reqs_for(n, seed0):
    out = []
    for j in range(n):
        r = c2_inputs(unet, seed=seed0 * 16 + j)
        r["region_masks"] = masks
        out.append(r)
    return out


def timed(fn, steps):
    fn(0)                                        # warm-up (captures the graphs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        fn(1 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


if a.what in ("ips", "both"):
    concept = make_concept_models(unet, n_concepts=2, rank=64)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    kw = dict(height=1024, width=1024, num_inference_steps=50, guidance_scale=7.5, cross_attention_kwargs={"scale": 0.8}, controller=ctl,
              concept_models=concept, lora_list=["concept0", "concept1"], styleL=False, use_graph=True)

    def stage2(reqs):
        ctl.reset()
        lat = pipe.generate_many(reqs, stage=2, **kw)
        for j in range(lat.shape[0]):
            vae.decode_latents(lat[j])
        return lat

    if a.what == "ips":
        for n in [int(x) for x in a.ips.split(",")]:
            sec = timed(lambda i: stage2(reqs_for(n, i)), a.steps)
            print(json.dumps({"measurement": "images_per_step", "images_per_step": n, "dtype": a.dtype, "images_per_sec": n / sec, "sec_per_step": sec,
                              "workload": "BASELINE configs[1], stage-2 call + VAE decode of both images", "steps_timed": a.steps}), flush=True)
    else:
        n = int(a.ips.split(",")[-1])
        def both(i):
            rq = reqs_for(n, i)
            ctl.reset()
            lat1 = pipe.generate_many([{k: v for k, v in r.items() if k != "region_masks"} for r in rq], stage=1, **kw)     # stage 1: 50 plain steps
            for j in range(n):
                vae.decode_latents(lat1[j])                                     # the stage-1 image the segmenter would look at
            stage2(rq)
        sec = timed(both, a.steps)
        print(json.dumps({"measurement": "stage1_plus_stage2", "images_per_step": n, "dtype": a.dtype, "images_per_sec": n / sec, "sec_per_step": sec,
                          "workload": "BASELINE configs[1]: stage 1 (50 plain steps, decode) + stage 2 (fusion for i > 15, decode) per image; "
                                      "detection / segmentation between the stages excluded; 3.626 PFLOP per image (SURVEY 8d)", "steps_timed": a.steps}), flush=True)
        # ---- the same two calls with the exact redundancies of SURVEY 7.4 taken out (round 4): stage 1 runs [unc, cond] once per request
        # (dedup), stage 2 resumes at the first fused step from the latents stage 1 left in the StageCache and keeps dedup off (nothing
        # left to deduplicate): 100 + 272 = 372 of the reference's 536 sample-forwards; latents compared bitwise with the full flow
        from omg_amd.pipeline import StageCache
        last = {}
        def both_dedup(i):
            rq = reqs_for(n, i)
            cache = StageCache()
            ctl.reset()
            lat1 = pipe.generate_many([{k: v for k, v in r.items() if k != "region_masks"} for r in rq], stage=1, dedup=True, stage_cache=cache, **kw)
            for j in range(n):
                vae.decode_latents(lat1[j])
            ctl.reset()
            lat2 = pipe.generate_many(rq, stage=2, stage_cache=cache, **kw)
            assert cache.hits == n, (cache.hits, cache.misses)
            for j in range(n):
                vae.decode_latents(lat2[j])
            last["lat"], last["i"] = lat2, i
        sec_d = timed(both_dedup, a.steps)
        drop = {"on": False}
        def both_min(i):      # + the base sample's stage-1 trajectory from the cache: the resumed steps run without `unc0` (338 forwards)
            rq = reqs_for(n, i)
            cache = StageCache()
            ctl.reset()
            lat1 = pipe.generate_many([{k: v for k, v in r.items() if k != "region_masks"} for r in rq], stage=1, dedup=True, stage_cache=cache, **kw)
            for j in range(n):
                vae.decode_latents(lat1[j])
            ctl.reset()
            lat2 = pipe.generate_many(rq, stage=2, stage_cache=cache, drop_unc0=True, **kw)
            for j in range(n):
                vae.decode_latents(lat2[j])
            last["lat_min"], last["i_min"] = lat2, i
        sec_m = timed(both_min, a.steps)
        ctl.reset()
        want = pipe.generate_many(reqs_for(n, last["i"]), stage=2, **kw)
        print(json.dumps({"measurement": "stage1_plus_stage2_dedup", "images_per_step": n, "dtype": a.dtype, "images_per_sec": n / sec_d, "sec_per_step": sec_d,
                          "sample_forwards_executed": 372, "sample_forwards_reference": 536, "latents_bitwise_equal_to_full_flow": bool(torch.equal(want, last["lat"])),
                          "workload": "the same flow with SURVEY 7.4's exact redundancies removed: stage 1 deduplicated ([unc, cond] once per request: 100 "
                                      "sample-forwards), stage 2 resumed at step 16 from the stage-1 latents (StageCache: 34 x 8 = 272); both decodes kept",
                          "steps_timed": a.steps}), flush=True)
        ctl.reset()
        want_m = pipe.generate_many(reqs_for(n, last["i_min"]), stage=2, **kw)
        print(json.dumps({"measurement": "stage1_plus_stage2_minimal", "images_per_step": n, "dtype": a.dtype, "images_per_sec": n / sec_m, "sec_per_step": sec_m,
                          "sample_forwards_executed": 338, "sample_forwards_reference": 536, "latents_bitwise_equal_to_full_flow": bool(torch.equal(want_m, last["lat_min"])),
                          "workload": "SURVEY 7.4 in full: stage 1 deduplicated (100), stage 2 resumed at step 16 WITHOUT unc0 — the base sample's latents come from the "
                                      "stage-1 trajectory in the StageCache (34 x 7 = 238); both decodes kept", "steps_timed": a.steps}), flush=True)
elif a.what == "config4":
    from omg_amd.controlnet import ControlNetModel
    cn = random_init_(ControlNetModel(UNetConfig.sdxl(), dtype=dt, device=dev), 7)
    if a.dtype == "fp8":      # round 4: BASELINE configs[4] says "fp8 MFMA" for the step — the ControlNet of the main pass included
        from omg_amd.unet import MX8_CLASSES
        cn.set_precision_classes(MX8_CLASSES)
    concept = make_concept_models(unet, n_concepts=3, rank=64, style=True)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    m3 = torch.zeros(1024, 1024, device=dev); m3[128:512, 300:700] = 1
    masks3 = masks + [m3]
    n = int(a.ips.split(",")[-1])
    pose = torch.rand(1, 3, 1024, 1024, generator=torch.Generator().manual_seed(9))

    def reqs4(i):
        out = []
        for j in range(n):
            r = c2_inputs(unet, seed=i * 16 + j, n_concepts=3)
            r["region_masks"] = masks3
            out.append(r)
        return out

    def run4(i):
        ctl.reset()
        lat = pipe.generate_many(reqs4(i), height=1024, width=1024, num_inference_steps=50, guidance_scale=7.5, cross_attention_kwargs={"scale": 0.8},
                                 controller=ctl, concept_models=concept, stage=2, lora_list=["concept0", "concept1", "concept2"], styleL=True,
                                 controlnet=cn, controlnet_image=pose, controlnet_conditioning_scale=1.0, use_graph=True)
        for j in range(n):
            vae.decode_latents(lat[j])
        assert torch.isfinite(lat).all()
    sec = timed(run4, a.steps)
    pf = 3.337
    print(json.dumps({"measurement": "config4", "images_per_step": n, "dtype": a.dtype, "images_per_sec": n / sec, "sec_per_step": sec,
                      "end_to_end_tflops": pf * 1e3 * n / sec,
                      "workload": "BASELINE configs[4]: SDXL 1024^2, 50 DDIM steps, ControlNet (openpose-sdxl architecture, 1.25 B parameters) on the main pass, 3 concepts "
                                  "with overlapping masks, style LoRA on the main pass and [0.7, 0.5] with each concept LoRA, stage-2 call + upcast VAE decode; "
                                  "3.337 PFLOP per image (SURVEY 8d); fp8 = MX-fp8 on the transformer Linears + resnet convolutions of the UNet AND (round 4) of the ControlNet",
                      "steps_timed": a.steps}), flush=True)
else:
    from omg_amd.controlnet import ControlNetModel
    from omg_amd.ip_adapter import IPAdapter
    idn = random_init_(ControlNetModel(UNetConfig.sdxl(), dtype=dt, device=dev), 5)
    IPAdapter(unet, num_tokens=16, scale=0.8).init_synthetic_(seed=3)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("euler"))
    concept = ConceptModels(unet, None)
    n = 8

    def reqs(i):
        out = []
        gg = torch.Generator().manual_seed(1000 + i)
        for j in range(n):
            r = c2_inputs(unet, seed=i * 16 + j)
            r["region_masks"] = masks
            r["region_image_embeds"] = [torch.randn(2, 16, 2048, generator=gg).to(dt).to(dev) for _ in range(2)]
            r["kps_image"] = torch.rand(1, 3, 1024, 1024, generator=gg)
            out.append(r)
        return out

    def run(i):
        ctl.reset()
        lat = pipe.generate_many(reqs(i), height=1024, width=1024, num_inference_steps=30, guidance_scale=3.0, controller=ctl, concept_models=concept,
                                 stage=2, lora_list=["id0", "id1"], styleL=False, identitynet=idn, identitynet_conditioning_scale=0.8, use_graph=True)
        for j in range(n):
            vae.decode_latents(lat[j])
    sec = timed(run, a.steps)
    print(json.dumps({"measurement": "config2_instantid", "images_per_step": n, "dtype": a.dtype, "images_per_sec": n / sec, "sec_per_step": sec,
                      "workload": "BASELINE configs[2]: OMG + InstantID, 2 identities, 1024^2, 30 Euler steps, guidance 3.0, IdentityNet on each concept pass "
                                  "(14 fused steps), IP-Adapter (16 face tokens), stage-2 call + VAE decode; 1.361 PFLOP per image (SURVEY 8d)",
                      "steps_timed": a.steps}), flush=True)
