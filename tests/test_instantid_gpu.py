"""Loop parity (-m gpu) for the ControlNet-conditioned flows:
  * LoRA pipeline with a ControlNet on the main pass (lora_pipeline.py:519-536, BASELINE config 5's structure);
  * OMG + InstantID (instantid_pipeline.py:574-683, BASELINE config 3's structure): IdentityNet ControlNet on the concept
    pass fed with face tokens + key-point image, concept UNet cross-attention = text + scale * image-prompt tokens.
Both against the oracle's literal loop with closures built from oracle/unet.py, oracle/controlnet.py, oracle/ip_adapter.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omg_amd import controller as pc
from omg_amd.controlnet import ControlNetModel
from omg_amd.ip_adapter import IPAdapter
from omg_amd.lora import LoraAdapter, LoraBank
from omg_amd.pipeline import ConceptModels, LoraMultiConceptPipeline, revise_regionally_controlnet_forward
from omg_amd.schedulers import make_scheduler
from omg_amd.unet import UNet2DConditionModel, UNetConfig
from oracle import controller as oc
from oracle import controlnet as ocn
from oracle import ip_adapter as oip
from oracle import pipeline as opipe
from oracle import schedulers as osched
from oracle import unet as ou

P = "a man and a woman walking on the street"
dtype = torch.float16


def emb(cfg, n, seed, tokens=77):
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(n, tokens, cfg.cross_attention_dim, generator=g).to(dtype).float()
    p = torch.randn(n, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g).to(dtype).float()
    return e, p


def common_setup(dev):
    cfg, ocfg = UNetConfig.tiny(), ou.UNetConfig.tiny()
    sd = ou.init_state_dict(ocfg, seed=0, dtype=dtype)
    unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev)
    unet.load_state_dict({k: v.to(dtype) for k, v in sd.items()})
    csd = ocn.init_state_dict(ocfg, seed=5, dtype=dtype)
    cn = ControlNetModel(cfg, dtype=dtype, device=dev)
    cn.load_state_dict({k: v.to(dtype) for k, v in csd.items()})
    return cfg, ocfg, sd, unet, csd, cn


@pytest.mark.parametrize("use_graph,style", [(False, False), (True, False), (False, True), (True, True)])
def test_instantid_loop_matches_oracle(dev, use_graph, style):
    """``style``: a LoRA that ``load_lora_weights`` left ACTIVE on both pipes (inference_instantid.py:220-222; the InstantID loop never
    calls ``set_adapters``): main rows at ``cross_attention_kwargs["scale"]`` = 0.8 (instantid_pipeline.py:596-616), concept rows at 1.0
    (``cross_attention_kwargs=None``, :665-674); the IdentityNet carries no LoRA."""
    cfg, ocfg, sd, unet, csd, idn = common_setup(dev)
    L = cfg.sample_size
    S, gs, fstart, ip_scale, idn_scale, ntok = 7, 3.0, 2, 0.8, 0.8, 16       # guidance 3.0 (inference_instantid.py:78)
    H = W = L * 8
    pos_e, pos_p = emb(cfg, 1, 2); neg_e, neg_p = emb(cfg, 1, 1)
    pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    regions, faces = [], []
    g = torch.Generator().manual_seed(77)
    for c in range(2):
        re_, rp_ = emb(cfg, 2, 10 + c)
        regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
        idtok = torch.randn(1, ntok, cfg.cross_attention_dim, generator=g).to(dtype).float()
        faces.append(torch.cat([torch.randn(1, ntok, cfg.cross_attention_dim, generator=g).to(dtype).float() * 0.1, idtok], dim=0))  # [zero-id proj, id]
    m1 = torch.zeros(H, W); m1[H // 4:, W // 16: W // 2] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 16:] = 1
    kps = torch.rand(1, 3, H, W, generator=g).to(dtype).float()
    lat0 = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(14))
    tid = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32)
    # IP-Adapter weights on every attn2
    ipw = {}
    gi = torch.Generator().manual_seed(9)
    for name, shp in ou.param_shapes(ocfg).items():
        if name.endswith(".attn2.to_k.weight"):
            mod = name[: -len(".to_k.weight")]
            c_, cx = shp
            ipw[mod] = ((torch.randn(c_, cx, generator=gi) * cx ** -0.5).to(dtype).float(), (torch.randn(c_, cx, generator=gi) * cx ** -0.5).to(dtype).float())
    ipa = IPAdapter(unet, num_tokens=ntok, scale=ip_scale)
    ipa.load_named(ipw)
    args = ([P, P], S, {"default_": 1.0}, 0.4, L // 4, L // 4)
    pctl = pc.AttentionReplace(*args, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("euler"))
    bank, lora_main, lora_conc, style_kw = None, None, None, {}
    if style:
        names = ou.lora_target_names(ocfg)
        w08, lora_main = ou.make_lora(ocfg, names, rank=8, seed=300, scale=0.8, dtype=dtype)
        w10, lora_conc = ou.make_lora(ocfg, names, rank=8, seed=300, scale=1.0, dtype=dtype)
        assert all(torch.equal(w08[k][0], w10[k][0]) and torch.equal(w08[k][1], w10[k][1]) for k in w08)      # one adapter, two scales
        bank = LoraBank(unet, [LoraAdapter("style", {k: (a.to(dev), b.to(dev)) for k, (a, b) in w08.items()})])
        style_kw = dict(main_adapters=[("style", 1.0)], concept_adapters=[("style", 1.0)], concept_adapter_scale=1.0,
                        cross_attention_kwargs={"scale": 0.8}, concept_lora=False)
    concept = ConceptModels(unet, bank)
    req = dict(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp,
               region_prompt_embeds=regions, region_masks=[m1, m2], latents=lat0, region_image_embeds=faces, kps_image=kps)
    pctl.reset()
    traj = []
    pipe.generate_many([req], height=H, width=W, num_inference_steps=S, guidance_scale=gs, controller=pctl, concept_models=concept,
                       stage=2, lora_list=["id0", "id1"], styleL=False, fusion_start=fstart, identitynet=idn,
                       identitynet_conditioning_scale=idn_scale, trajectory=traj, use_graph=use_graph, **style_kw)
    if use_graph:
        # a SECOND request with other identities through the captured graphs must equal its own eager run bit for bit: a replay runs
        # no Python, so every cached projection of per-request inputs (text K/V, IdentityNet K/V and conditioning, the UNet's
        # image-prompt K/V) has to be refreshed eagerly first (round 3: the last one was not — the first request's faces came back)
        g2 = torch.Generator().manual_seed(78)
        faces2 = [torch.cat([torch.randn(1, ntok, cfg.cross_attention_dim, generator=g2).to(dtype).float() * 0.1,
                             torch.randn(1, ntok, cfg.cross_attention_dim, generator=g2).to(dtype).float()], dim=0) for _ in range(2)]
        req2 = dict(req, region_image_embeds=faces2, kps_image=torch.rand(1, 3, H, W, generator=g2).to(dtype).float(),
                    latents=torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(15)))
        kw2 = dict(height=H, width=W, num_inference_steps=S, guidance_scale=gs, controller=pctl, concept_models=concept, stage=2,
                   lora_list=["id0", "id1"], styleL=False, fusion_start=fstart, identitynet=idn, identitynet_conditioning_scale=idn_scale, **style_kw)
        pctl.reset()
        second_graph = pipe.generate_many([req2], use_graph=True, **kw2)
        pctl.reset()
        second_eager = pipe.generate_many([req2], use_graph=False, **kw2)
        assert torch.equal(second_graph, second_eager), (second_graph - second_eager).abs().max()
        pctl.reset()
        first_again = pipe.generate_many([req], use_graph=True, **kw2)
        assert torch.equal(first_again[0].float().cpu(), traj[-1][0].float().cpu())
    # ---- oracle
    osch = osched.make("euler", S)
    octl = oc.AttentionReplaceOracle(*args)
    octl.num_att_layers = pctl.num_att_layers
    attn_main = oc.reference_attn_fn(octl)
    ip_fn = oip.make_ip_attn_fn(ipw, ip_scale, ntok)
    ctx4, te4 = torch.cat([ne, pe]), torch.cat([npp, pp])

    def main(x, i):
        return ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx4, te4, tid.repeat(4, 1), attn_fn=attn_main, lora=lora_main)

    def conc(c):
        ctx2 = torch.cat([regions[c][0], regions[c][1]]); te2 = torch.cat([regions[c][2], regions[c][3]])
        def f(x, i):
            t = float(osch.timesteps[i])
            down, mid = ocn.controlnet_forward(csd, ocfg, x, t, faces[c], kps.repeat(2, 1, 1, 1), idn_scale, te2, tid.repeat(2, 1))
            return ou.unet_forward(sd, ocfg, x, t, torch.cat([ctx2, faces[c]], dim=1), te2, tid.repeat(2, 1), attn_fn=ip_fn,
                                   down_block_additional_residuals=down, mid_block_additional_residual=mid, lora=lora_conc)
        return f

    rec = []
    ref = opipe.denoise(main, [conc(0), conc(1)], osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=[m1, m2], fusion_start=fstart, record=rec)
    errs = [(a[0].float().cpu() - b).abs().max().item() for a, b in zip(traj, rec)]
    rel = errs[-1] / ref.pow(2).mean().sqrt().item()
    print(f"instantid (graph={use_graph}, style={style}): per-step max|d| = " + " ".join(f"{e:.2e}" for e in errs), f" rel {rel:.2e}")
    assert rel < 2e-2, errs
    # the identity branch must matter
    octl.reset()
    plain = opipe.denoise(main, [lambda x, i, c=c: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), torch.cat([regions[c][0], regions[c][1]]),
                                                                   torch.cat([regions[c][2], regions[c][3]]), tid.repeat(2, 1)) for c in range(2)],
                          osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=[m1, m2], fusion_start=fstart)
    assert (plain[1] - ref[1]).abs().max() > 0.1
    if style:      # ... and so must the adapter: the same call without it lands elsewhere
        pctl.reset()
        kw0 = {k: v for k, v in style_kw.items() if k in ("cross_attention_kwargs",)}
        no_style = pipe.generate_many([req], height=H, width=W, num_inference_steps=S, guidance_scale=gs, controller=pctl, concept_models=concept,
                                      stage=2, lora_list=["id0", "id1"], styleL=False, fusion_start=fstart, identitynet=idn,
                                      identitynet_conditioning_scale=idn_scale, concept_lora=False, **kw0)
        assert (no_style[0].float().cpu() - traj[-1][0].float().cpu()).abs().max() > 0.05


def test_lora_pipeline_with_main_controlnet(dev):
    cfg, ocfg, sd, unet, csd, cn = common_setup(dev)
    L = cfg.sample_size
    S, gs, fstart, cs = 6, 7.5, 2, 0.7
    H = W = L * 8
    pos_e, pos_p = emb(cfg, 1, 2); neg_e, neg_p = emb(cfg, 1, 1)
    pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    re_, rp_ = emb(cfg, 2, 10)
    regions = [(re_[0:1], re_[1:2], rp_[0:1], rp_[1:2])]
    m1 = torch.zeros(H, W); m1[H // 4:, W // 8: W // 2 + 8] = 1
    g = torch.Generator().manual_seed(3)
    pose = torch.rand(1, 3, H, W, generator=g).to(dtype).float()
    lat0 = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(14))
    tid = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32)
    names = ou.lora_target_names(ocfg)
    ow, olora = ou.make_lora(ocfg, names, rank=8, seed=100, scale=0.8, dtype=dtype)
    concept = ConceptModels(unet, LoraBank(unet, [LoraAdapter("c0", {k: (a.to(dev), b.to(dev)) for k, (a, b) in ow.items()})]))
    args = ([P, P], S, {"default_": 1.0}, 0.4, L // 4, L // 4)
    pctl = pc.AttentionReplace(*args, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    req = dict(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp,
               region_prompt_embeds=regions, region_masks=[m1], latents=lat0)
    outs = {}
    for use_graph in (False, True):
        pctl.reset()
        traj = []
        pipe.generate_many([req], height=H, width=W, num_inference_steps=S, guidance_scale=gs, cross_attention_kwargs={"scale": 0.8},
                           controller=pctl, concept_models=concept, stage=2, lora_list=["c0"], styleL=False, fusion_start=fstart,
                           controlnet=cn, controlnet_image=pose, controlnet_conditioning_scale=cs, trajectory=traj, use_graph=use_graph)
        outs[use_graph] = [t[0].cpu() for t in traj]
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
    osch = osched.make("ddim", S)
    octl = oc.AttentionReplaceOracle(*args)
    octl.num_att_layers = pctl.num_att_layers
    attn_main = oc.reference_attn_fn(octl)
    ctx4, te4 = torch.cat([ne, pe]), torch.cat([npp, pp])

    def main(x, i):
        t = float(osch.timesteps[i])
        down, mid = ocn.controlnet_forward(csd, ocfg, x, t, ctx4, pose.repeat(4, 1, 1, 1), cs, te4, tid.repeat(4, 1))
        return ou.unet_forward(sd, ocfg, x, t, ctx4, te4, tid.repeat(4, 1), attn_fn=attn_main, down_block_additional_residuals=down,
                               mid_block_additional_residual=mid)

    ctx2 = torch.cat([regions[0][0], regions[0][1]]); te2 = torch.cat([regions[0][2], regions[0][3]])
    conc = lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx2, te2, tid.repeat(2, 1), lora=olora)
    rec = []
    ref = opipe.denoise(main, [conc], osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=[m1], fusion_start=fstart, record=rec)
    errs = [(a.float() - b).abs().max().item() for a, b in zip(outs[False], rec)]
    rel = errs[-1] / ref.pow(2).mean().sqrt().item()
    print("lora+controlnet: per-step max|d| = " + " ".join(f"{e:.2e}" for e in errs), f" rel {rel:.2e}")
    assert rel < 2e-2, errs


def test_two_graph_mode_instantid_engines_of_different_shapes_alternate(dev):
    """the image-prompt K / V^T cache of a layer held ONE entry; engine A (one request), engine B (two requests: another
    token shape), engine A again re-allocated A's projections while A's captured graphs still read the freed tensors.  Now one entry per
    shape: the third call must reproduce the first bit for bit, and an eager run of the same request must agree with it."""
    cfg, ocfg, sd, unet, csd, idn = common_setup(dev)
    L = cfg.sample_size
    S, gs, fstart, ntok = 5, 3.0, 1, 16
    H = W = L * 8
    g = torch.Generator().manual_seed(5)

    def request(seed):
        gg = torch.Generator().manual_seed(seed)
        pos_e, pos_p = emb(cfg, 1, seed); neg_e, neg_p = emb(cfg, 1, seed + 1)
        regions, faces = [], []
        for c in range(2):
            re_, rp_ = emb(cfg, 2, seed + 10 + c)
            regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
            faces.append(torch.randn(2, ntok, cfg.cross_attention_dim, generator=gg).to(dtype).float())
        m1 = torch.zeros(H, W); m1[H // 4:, W // 16: W // 2] = 1
        m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 16:] = 1
        return dict(prompt_embeds=pos_e.repeat(2, 1, 1), negative_prompt_embeds=neg_e.repeat(2, 1, 1), pooled_prompt_embeds=pos_p.repeat(2, 1),
                    negative_pooled_prompt_embeds=neg_p.repeat(2, 1), region_prompt_embeds=regions, region_masks=[m1, m2],
                    latents=torch.randn(1, 4, L, L, generator=gg), region_image_embeds=faces, kps_image=torch.rand(1, 3, H, W, generator=gg).to(dtype).float())

    ipw = {}
    for name, shp in ou.param_shapes(ocfg).items():
        if name.endswith(".attn2.to_k.weight"):
            c_, cx = shp
            ipw[name[: -len(".to_k.weight")]] = ((torch.randn(c_, cx, generator=g) * cx ** -0.5).to(dtype).float(), (torch.randn(c_, cx, generator=g) * cx ** -0.5).to(dtype).float())
    IPAdapter(unet, num_tokens=ntok, scale=0.8).load_named(ipw)
    pctl = pc.AttentionReplace([P, P], S, {"default_": 1.0}, 0.4, L // 4, L // 4, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("euler"))
    concept = ConceptModels(unet, None)
    kw = dict(height=H, width=W, num_inference_steps=S, guidance_scale=gs, controller=pctl, concept_models=concept, stage=2,
              lora_list=["id0", "id1"], styleL=False, fusion_start=fstart, identitynet=idn, identitynet_conditioning_scale=0.8)
    ra, rb1, rb2 = request(100), request(200), request(300)

    def run(reqs, graph):
        pctl.reset()
        return pipe.generate_many(reqs, use_graph=graph, **kw).clone()

    a1 = run([ra], True)
    b1 = run([rb1, rb2], True)
    a2 = run([ra], True)                    # A's graphs replay against A's own K / V^T tensors
    b2 = run([rb1, rb2], True)
    assert torch.equal(a1, a2) and torch.equal(b1, b2)
    assert torch.equal(a1, run([ra], False)), "graph replay == eager"
    assert torch.equal(b1[0:1], run([rb1], True)), "batched == single (a third token shape in the cache)"
