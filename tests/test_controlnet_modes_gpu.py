"""The ControlNet call modes of the reference's loop that round 5 refused and round 6 implements (-m gpu; VERDICT r5 missing 3 / next 7):

  * ``control_guidance_start`` / ``control_guidance_end`` -> ``controlnet_keep[i]`` (lora_pipeline.py:275-286, :421-428, :511-517),
  * ``guess_mode=True`` (:497-503, :531-535): the nets see the conditional rows only (diffusers' logspace residual scaling), zeros for the rest,
  * a LIST of ControlNets (MultiControlNetModel, :175-176, :366-383, :511-512): one image / scale / window per net, residuals summed,
  * the InstantID twin of the window (instantid_pipeline.py:477-483, :566-578): the one factor scales the IdentityNet and the t2i ControlNet.

HIP pipeline (fp16) vs the oracle's loop on the same fp16-rounded weights; the oracle functions used here — ``oracle/pipeline.controlnet_keep`` and
``main_controlnet_residuals`` — are pinned by the reference's own loop run in the build container (tests/golden/make_golden_loop.py cases
``ddim_controlnet_window`` / ``_guess`` / ``euler_controlnet_multi``, tests/test_oracle_loop.py).  hipGraph replay must equal the eager loop bitwise
(a step's conditioning scales are part of the captured regime)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omg_amd import controller as pc
from omg_amd.controlnet import ControlNetModel
from omg_amd.lora import LoraAdapter, LoraBank
from omg_amd.pipeline import ConceptModels, LoraMultiConceptPipeline, controlnet_keep, revise_regionally_controlnet_forward
from omg_amd.schedulers import make_scheduler
from omg_amd.unet import UNet2DConditionModel, UNetConfig
from oracle import controller as oc
from oracle import controlnet as ocn
from oracle import pipeline as opipe
from oracle import schedulers as osched
from oracle import unet as ou

P = "a man and a woman walking on the street"
dtype = torch.float16

VARIANTS = {
    "window": dict(control_guidance_start=0.25, control_guidance_end=0.75),
    "guess": dict(guess_mode=True),
    "guess_window": dict(guess_mode=True, control_guidance_start=0.0, control_guidance_end=0.5),
    "multi": dict(multi=True, control_guidance_start=[0.0, 0.25], control_guidance_end=[0.5, 1.0]),
    "multi_guess": dict(multi=True, guess_mode=True),
}


def emb(cfg, n, seed):
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(n, 77, cfg.cross_attention_dim, generator=g).to(dtype).float()
    p = torch.randn(n, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g).to(dtype).float()
    return e, p


# the two COMBINATIONS run under OMG_RUN_SLOW=1 only (25 s each; the driver's `-m gpu` budget): each mode on its own is in the default run
@pytest.mark.parametrize("variant", [v if v in ("window", "guess", "multi") else pytest.param(v, marks=pytest.mark.slow) for v in VARIANTS])
def test_controlnet_window_guess_mode_and_lists_match_the_oracle_loop(dev, variant):
    kw = dict(VARIANTS[variant])
    multi, guess = kw.pop("multi", False), kw.get("guess_mode", False)
    cfg, ocfg = UNetConfig.tiny(), ou.UNetConfig.tiny()
    sd = ou.init_state_dict(ocfg, seed=0, dtype=dtype)
    unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev)
    unet.load_state_dict({k: v.to(dtype) for k, v in sd.items()})
    csds, cns = [], []
    for seed in ((3, 4) if multi else (3,)):
        csd = ocn.init_state_dict(ocfg, seed=seed, dtype=dtype)
        cn = ControlNetModel(cfg, dtype=dtype, device=dev)
        cn.load_state_dict({k: v.to(dtype) for k, v in csd.items()})
        csds.append(csd); cns.append(cn)
    L = cfg.sample_size
    S, gs, fstart = 8, 7.5, 3
    H = W = L * 8
    pos_e, pos_p = emb(cfg, 1, 2); neg_e, neg_p = emb(cfg, 1, 1)
    pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    regions = []
    for c in range(2):
        re_, rp_ = emb(cfg, 2, 10 + c)
        regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
    m0 = torch.zeros(H, W); m0[H // 4:, W // 16: W // 2] = 1
    m1 = torch.zeros(H, W); m1[H // 4:, W // 2 - 16: W - 8] = 1
    masks = [m0, m1]
    g = torch.Generator().manual_seed(3)
    poses = [torch.rand(1, 3, H, W, generator=g).to(dtype).float() for _ in cns]
    scales = [0.7, 0.6][: len(cns)]
    lat0 = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(14))
    tid = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32)
    names = ou.lora_target_names(ocfg)
    ow, ofn = {}, {}
    for nm_, seed in (("c0", 100), ("c1", 101)):
        ow[nm_], ofn[nm_] = ou.make_lora(ocfg, names, rank=8, seed=seed, scale=0.8, dtype=dtype)
    concept = ConceptModels(unet, LoraBank(unet, [LoraAdapter(k, {n: (a.to(dev), b.to(dev)) for n, (a, b) in w.items()}) for k, w in ow.items()]))
    args = ([P, P], S, {"default_": 1.0}, 0.4, L // 4, L // 4)
    pctl = pc.AttentionReplace(*args, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    pipe.controlnet = cns if multi else cns[0]

    def run(use_graph=False, n_req=1):
        pctl.reset()
        traj = []
        common = dict(output_type="latent", height=H, width=W, num_inference_steps=S, guidance_scale=gs, cross_attention_kwargs={"scale": 0.8}, controller=pctl,
                      concept_models=concept, stage=2, lora_list=["c0", "c1"], styleL=False, fusion_start=fstart, use_graph=use_graph, trajectory=traj,
                      controlnet_conditioning_scale=scales if multi else scales[0], **kw)
        if n_req == 1:      # the reference's call signature (B3), the list of images as the reference takes it (lora_pipeline.py:366-383)
            pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, latents=lat0,
                 region_masks=masks, region_prompt_embeds=regions, image=[p_.to(dev) for p_ in poses] if multi else poses[0].to(dev), **common)
            return torch.stack([t.cpu() for t in traj])
        req = dict(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, region_prompt_embeds=regions,
                   region_masks=masks, latents=lat0)
        req2 = dict(req, latents=lat0.flip(-1).contiguous())
        for k_ in ("output_type", "trajectory"):
            common.pop(k_)
        lat = pipe.generate_many([req, req2], controlnet=pipe.controlnet, controlnet_image=[p_.to(dev) for p_ in poses] if multi else poses[0].to(dev), **common)
        return lat.cpu()

    # ---- the oracle's literal loop
    osch = osched.make("ddim", S)
    octl = oc.AttentionReplaceOracle(*args)
    octl.num_att_layers = pctl.num_att_layers
    attn_main = oc.reference_attn_fn(octl)
    ctx4, te4 = torch.cat([ne, pe]), torch.cat([npp, pp])
    keep = opipe.controlnet_keep(S, kw.get("control_guidance_start", 0.0), kw.get("control_guidance_end", 1.0), len(cns))
    assert keep == controlnet_keep(S, kw.get("control_guidance_start", 0.0), kw.get("control_guidance_end", 1.0), len(cns))
    nets = [lambda x, i, ctx, img, sc, te, tid_, gm, csd=csd: ocn.controlnet_forward(csd, ocfg, x, float(osch.timesteps[i]), ctx, img, sc, te, tid_, guess_mode=gm)
            for csd in csds]

    def main(x, i):
        down, mid = opipe.main_controlnet_residuals(nets, x, i, ctx4, te4, tid.repeat(4, 1), poses, scales, keep[i], guess)
        return ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx4, te4, tid.repeat(4, 1), attn_fn=attn_main,
                               down_block_additional_residuals=down, mid_block_additional_residual=mid)

    def conc(c):
        ctx2 = torch.cat([regions[c][0], regions[c][1]]); te2 = torch.cat([regions[c][2], regions[c][3]])
        return lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx2, te2, tid.repeat(2, 1), lora=ofn[f"c{c}"])

    rec = []
    opipe.denoise(main, [conc(0), conc(1)], osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=masks, fusion_start=fstart, record=rec)
    ref = torch.stack(rec)
    rms = ref[-1].pow(2).mean().sqrt().item()
    # the mode must matter: without the ControlNet(s) the trajectory is elsewhere
    rec0 = []
    octl.reset()
    opipe.denoise(lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx4, te4, tid.repeat(4, 1), attn_fn=attn_main), [conc(0), conc(1)], osch,
                  lat0 * osch.init_noise_sigma, S, gs, 2, masks=masks, fusion_start=fstart, record=rec0)
    assert (ref[-1] - rec0[-1]).abs().max().item() / rms > 0.05

    got = run()
    errs = [(a - b).abs().max().item() / rms for a, b in zip(got, ref)]
    print(f"{variant}: per-step max|d| / latent rms = " + " ".join(f"{e:.2e}" for e in errs))
    assert errs[-1] < 2e-2, errs
    # hipGraph replay: every distinct tuple of conditioning scales is its own captured regime; two passes so that replays (not only captures) run
    g1 = run(use_graph=True)
    g2 = run(use_graph=True)
    assert torch.equal(g1, got) and torch.equal(g2, got)
    # two requests in lock-step: request 0 must be the single-request call, bit for bit (the guess-mode residuals address rows per request)
    two = run(n_req=2)
    assert torch.equal(two[0], got[-1])
    assert not torch.equal(two[1], got[-1])


def test_instantid_guidance_window_scales_identitynet_and_t2i(dev):
    """instantid_pipeline.py:477-483, :566-578: `controlnet_keep[i]` multiplies BOTH the IdentityNet's scale (concept rows) and the t2i ControlNet's (main
    rows).  A window that closes both nets for every step must equal the call without them; a window over every step must equal the default call."""
    from omg_amd.ip_adapter import IPAdapter
    from omg_amd.pipeline import InstantidMultiConceptPipeline
    cfg, ocfg = UNetConfig.tiny(), ou.UNetConfig.tiny()
    sd = ou.init_state_dict(ocfg, seed=0, dtype=dtype)
    unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev)
    unet.load_state_dict({k: v.to(dtype) for k, v in sd.items()})
    nets = []
    for seed in (3, 4):
        cn = ControlNetModel(cfg, dtype=dtype, device=dev)
        cn.load_state_dict({k: v.to(dtype) for k, v in ocn.init_state_dict(ocfg, seed=seed, dtype=dtype).items()})
        nets.append(cn)
    IPAdapter(unet, num_tokens=16, scale=0.8).init_synthetic_(seed=9)
    g = torch.Generator().manual_seed(9)
    L = cfg.sample_size
    S, gs, fstart = 6, 3.0, 2
    H = W = L * 8
    pos_e, pos_p = emb(cfg, 1, 2); neg_e, neg_p = emb(cfg, 1, 1)
    pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    re_, rp_ = emb(cfg, 2, 10)
    regions = [(re_[0:1], re_[1:2], rp_[0:1], rp_[1:2])]
    tokens = [torch.randn(2, 16, cfg.cross_attention_dim, generator=g).to(dtype)]
    m0 = torch.zeros(H, W); m0[H // 4:, W // 8: W // 2] = 1
    kps, t2i = torch.rand(1, 3, H, W, generator=g), torch.rand(1, 3, H, W, generator=g)
    lat0 = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(14))
    pctl = pc.AttentionReplace([P, P], S, {"default_": 1.0}, 0.4, L // 4, L // 4, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = InstantidMultiConceptPipeline(unet, nets[0], make_scheduler("euler"), controlnet2=nets[1])

    def run(**kw):
        pctl.reset()
        return pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, image=kw.pop("image", kps.to(dev)),
                    t2i_image=kw.pop("t2i_image", t2i.to(dev)), height=H, width=W, num_inference_steps=S, guidance_scale=gs, latents=lat0,
                    controlnet_conditioning_scale=0.8, t2i_controlnet_conditioning_scale=0.6, controller=pctl, stage=2, region_masks=[m0],
                    region_prompt_embeds=regions, region_image_embeds=tokens, output_type="latent", fusion_start=fstart, **kw).images

    base = run()
    assert torch.equal(run(control_guidance_start=0.0, control_guidance_end=1.0), base)
    closed = run(control_guidance_start=0.99, control_guidance_end=1.0)          # i / S < 0.99 for every step: keep == 0 throughout
    assert not torch.equal(closed, base)
    # both nets silent: the t2i net adds nothing to the main rows and the IdentityNet nothing to the concept rows — but the concept UNet still sees the face
    # tokens through the IP-Adapter (instantid_pipeline.py:659-674), so the comparison is the same call with zero conditioning scales
    zero = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, image=kps.to(dev), t2i_image=t2i.to(dev),
                height=H, width=W, num_inference_steps=S, guidance_scale=gs, latents=lat0, controlnet_conditioning_scale=0.0, t2i_controlnet_conditioning_scale=0.0,
                controller=(pctl.reset(), pctl)[1], stage=2, region_masks=[m0], region_prompt_embeds=regions, region_image_embeds=tokens, output_type="latent",
                fusion_start=fstart).images
    assert (closed.float() - zero.float()).abs().max().item() < 1e-3 * zero.float().pow(2).mean().sqrt().item()
    half = run(control_guidance_start=0.0, control_guidance_end=0.5)
    assert not torch.equal(half, base) and not torch.equal(half, closed)
    assert torch.equal(run(control_guidance_start=0.0, control_guidance_end=0.5, use_graph=True), half)
