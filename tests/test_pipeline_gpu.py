"""Loop parity (-m gpu): omg_amd.pipeline.LoraMultiConceptPipeline (HIP) vs the oracle's literal loop
(oracle/pipeline.py) on the tiny SDXL-topology UNet: stage 1, and stage 2 with overlapping masks, a None
mask, per-concept LoRA and the p2p controller; per-step error growth is printed."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omg_amd import controller as pc
from omg_amd.lora import LoraAdapter, LoraBank
from omg_amd.pipeline import ConceptModels, LoraMultiConceptPipeline, revise_regionally_controlnet_forward
from omg_amd.schedulers import make_scheduler
from omg_amd.unet import UNet2DConditionModel, UNetConfig
from oracle import controller as oc
from oracle import pipeline as opipe
from oracle import schedulers as osched
from oracle import unet as ou

P = "a man and a woman walking on the street"


def setup(dev, dtype, seed=0):
    cfg, ocfg = UNetConfig.tiny(), ou.UNetConfig.tiny()
    sd = ou.init_state_dict(ocfg, seed=seed, dtype=dtype)
    unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev)
    unet.load_state_dict({k: v.to(dtype) for k, v in sd.items()})
    return cfg, ocfg, sd, unet


def embeds(cfg, n, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(n, 77, cfg.cross_attention_dim, generator=g).to(dtype).float()
    p = torch.randn(n, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g).to(dtype).float()
    return e, p


@pytest.mark.parametrize("sched,lora_mode,lh,lw", [("ddim", "merged", 16, 16), ("euler", "merged", 16, 16), ("ddim", "segment", 16, 16),
                                                   ("ddim", "merged", 24, 16), ("euler", "segment", 12, 20)])
def test_two_stage_loop_matches_oracle(dev, sched, lora_mode, lh, lw):
    """(lh, lw) != (16, 16): B3's `height` / `width` (lora_pipeline.py:217-218, :387-388) — non-square latents (the gradio buckets of the reference
    use them), ragged token counts (96 / 24, 60 / 15 tokens per level), the controller's `width * height` self-replace threshold
    (p2p_attention.py:114-118) with width != height."""
    dtype = torch.float16
    cfg, ocfg, sd, unet = setup(dev, dtype)
    L = cfg.sample_size
    S, gs, fstart = 8, 7.5, 3                      # fusion fires for i > 3 (the reference's 15, scaled to 8 steps)
    H, W = lh * 8, lw * 8
    neg_e, neg_p = embeds(cfg, 1, 1, dtype)
    pos_e, pos_p = embeds(cfg, 1, 2, dtype)
    pe, ne = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1)         # global prompt [p, p]
    pp, npp = pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    regions = []
    for c in range(3):
        re_, rp_ = embeds(cfg, 2, 10 + c, dtype)                  # [neg, pos] per concept
        regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
    m1 = torch.zeros(H, W); m1[H // 4:, W // 16: W // 2 - 8] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 24: W - 8] = 1    # overlaps m1 on purpose (sum rule)
    masks = [m1, None, m2]
    lat0 = torch.randn(1, 4, lh, lw, generator=torch.Generator().manual_seed(14))
    tid = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32)
    # synthetic LoRA adapters (attention + FF linears), rank 8, LoRA scale 0.8
    names = ou.lora_target_names(ocfg)
    ow, olora = [], []
    for c in range(3):
        w, fn = ou.make_lora(ocfg, names, rank=8, seed=100 + c, scale=0.8, dtype=dtype)
        ow.append(w); olora.append(fn)
    bank = LoraBank(unet, [LoraAdapter(f"c{c}", {k: (a.to(dev), b.to(dev)) for k, (a, b) in ow[c].items()}) for c in range(3)])
    concept = ConceptModels(unet, bank)
    args = ([P, P], S, {"default_": 1.0}, 0.4, lw // 4, lh // 4)          # width, height = the scripts' width // 32, height // 32 (inference_lora.py:247)
    pctl = pc.AttentionReplace(*args, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler(sched))
    osch = osched.make(sched, S)

    def oracle_run(stage):
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
        out = opipe.denoise(main, [conc(c) for c in range(3)], osch, lat0 * osch.init_noise_sigma, S, gs, stage,
                            masks=masks, fusion_start=fstart, record=rec)
        return out, rec

    for stage in (1, 2):
        pctl.reset()
        traj = []
        out = pipe(output_type="latent", prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp,
                   height=H, width=W, num_inference_steps=S, guidance_scale=gs, latents=lat0,
                   cross_attention_kwargs={"scale": 0.8}, controller=pctl, concept_models=concept, stage=stage,
                   region_masks=masks, lora_list=["c0", "c1", "c2"], styleL=False, region_prompt_embeds=regions,
                   trajectory=traj, fusion_start=fstart, lora_mode=lora_mode).images
        ref, rec = oracle_run(stage)
        errs = [(a.float().cpu() - b).abs().max().item() for a, b in zip(traj, rec)]
        print(f"{sched}/{lora_mode} {lh}x{lw} stage {stage}: per-step max|d| = " + " ".join(f"{e:.2e}" for e in errs), " latent rms", ref.pow(2).mean().sqrt().item())
        # fp16 noise-prediction error (~4e-3) is amplified by CFG (x16 at gs 7.5) and the scheduler's eps
        # coefficient every step; a logic error (mask, ordering, coefficient) would be O(latent rms)
        rel = errs[-1] / ref.pow(2).mean().sqrt().item()
        assert rel < 2e-2, (rel, errs)
        assert (pctl.cur_step, pctl.cur_att_layer) == (S, 0)
        if stage == 1:
            assert torch.equal(out[0], out[1]), "stage 1: both samples are identical (same latents, same prompt)"
            stage1 = ref
        else:
            assert (ref[1] - stage1[1]).abs().max() > 0.1, "fusion must change the edited sample"
            assert torch.allclose(ref[0], stage1[0], atol=1e-5), "the base sample never depends on the edit"


def test_graph_replay_is_bitwise_equal_to_eager(dev):
    """hipGraph path: static buffers + captured step graphs (3 regimes) vs the eager loop, two different images
    through the SAME engine (second image replays every step)."""
    dtype = torch.float16
    cfg, ocfg, sd, unet = setup(dev, dtype)
    L = cfg.sample_size
    S, gs, fstart = 10, 7.5, 3
    H = W = L * 8
    names = ou.lora_target_names(ocfg)
    bank = LoraBank(unet, [LoraAdapter(nm, {k: (a.to(dev), b.to(dev)) for k, (a, b) in ou.make_lora(ocfg, names, 8, 100 + c, 0.8, dtype)[0].items()})
                           for c, nm in enumerate(["c0", "c1", "style"])])
    concept = ConceptModels(unet, bank)
    pctl = pc.AttentionReplace([P, P], S, {"default_": 1.0}, 0.5, L // 4, L // 4, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    m1 = torch.zeros(H, W); m1[H // 4:, : W // 2] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 16:] = 1

    def run(seed, use_graph):
        pe1, pp1 = embeds(cfg, 1, seed, dtype); ne1, np1 = embeds(cfg, 1, seed + 50, dtype)
        regions = []
        for c in range(2):
            re_, rp_ = embeds(cfg, 2, seed + 10 + c, dtype)
            regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
        lat0 = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(seed))
        pctl.reset()
        traj = []
        pipe(output_type="latent", prompt_embeds=pe1.repeat(2, 1, 1), negative_prompt_embeds=ne1.repeat(2, 1, 1), pooled_prompt_embeds=pp1.repeat(2, 1),
             negative_pooled_prompt_embeds=np1.repeat(2, 1), height=H, width=W, num_inference_steps=S, guidance_scale=gs, latents=lat0,
             cross_attention_kwargs={"scale": 0.8}, controller=pctl, concept_models=concept, stage=2, region_masks=[m1, m2],
             lora_list=["c0", "c1"], styleL=False, region_prompt_embeds=regions, trajectory=traj, fusion_start=fstart, use_graph=use_graph)
        assert (pctl.cur_step, pctl.cur_att_layer) == (S, 0)
        return [t.cpu() for t in traj]

    eager = [run(seed, False) for seed in (1, 2)]
    graph = [run(seed, True) for seed in (1, 2)]       # image 1 captures, image 2 is pure replay
    graph.append(run(1, True))                          # and back to image 1 through the same graphs
    for a, b in ((eager[0], graph[0]), (eager[1], graph[1]), (eager[0], graph[2])):
        for i, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), f"step {i}: graph replay differs from eager by {(x - y).abs().max().item()}"
    assert not torch.equal(eager[0][-1], eager[1][-1])


def test_generate_many_equals_single_requests(dev):
    """Two requests in lock-step through one batched forward per step == the same requests run one at a time
    (bitwise: every kernel is batch-invariant)."""
    dtype = torch.float16
    cfg, ocfg, sd, unet = setup(dev, dtype)
    L = cfg.sample_size
    S, gs, fstart = 6, 7.5, 2
    H = W = L * 8
    names = ou.lora_target_names(ocfg)
    bank = LoraBank(unet, [LoraAdapter(nm, {k: (a.to(dev), b.to(dev)) for k, (a, b) in ou.make_lora(ocfg, names, 8, 100 + c, 0.8, dtype)[0].items()})
                           for c, nm in enumerate(["c0", "c1", "style"])])
    concept = ConceptModels(unet, bank)
    pctl = pc.AttentionReplace([P, P], S, {"default_": 1.0}, 0.5, L // 4, L // 4, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))

    def request(seed):
        pe1, pp1 = embeds(cfg, 1, seed, dtype); ne1, np1 = embeds(cfg, 1, seed + 50, dtype)
        regions = []
        for c in range(2):
            re_, rp_ = embeds(cfg, 2, seed + 10 + c, dtype)
            regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
        m1 = torch.zeros(H, W); m1[H // 4:, : W // 2 + 8 * seed] = 1
        m2 = torch.zeros(H, W); m2[H // 8 * seed:, W // 2 - 16:] = 1
        return dict(prompt_embeds=pe1.repeat(2, 1, 1), negative_prompt_embeds=ne1.repeat(2, 1, 1), pooled_prompt_embeds=pp1.repeat(2, 1),
                    negative_pooled_prompt_embeds=np1.repeat(2, 1), region_prompt_embeds=regions, region_masks=[m1, m2],
                    latents=torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(seed)))

    common = dict(height=H, width=W, num_inference_steps=S, guidance_scale=gs, cross_attention_kwargs={"scale": 0.8}, controller=pctl,
                  concept_models=concept, stage=2, lora_list=["c0", "c1"], styleL=False, fusion_start=fstart)
    singles = []
    for seed in (1, 2, 3):
        pctl.reset()
        singles.append(pipe.generate_many([request(seed)], **common)[0].cpu())
    for use_graph in (False, True):
        pctl.reset()
        many = pipe.generate_many([request(1), request(2), request(3)], use_graph=use_graph, **common).cpu()
        assert (pctl.cur_step, pctl.cur_att_layer) == (S, 0)
        for j in range(3):
            assert torch.equal(many[j], singles[j]), f"request {j} (graph={use_graph}): max diff {(many[j] - singles[j]).abs().max().item()}"


@pytest.mark.parametrize("styleL", [False, True])
def test_dedup_of_identical_samples_is_bitwise_equal_to_the_full_batch(dev, styleL):
    """SURVEY §7.4 / VERDICT r2 item 5: until the first fused step (stage 1: always) samples 0 and 1 of a call are the same
    computation; ``dedup=True`` runs [unc, cond] once per request in those steps.  Every step's latents must equal the full
    [unc0, unc1, cond0, cond1] run bit for bit — eager and through captured graphs, one request and two in lock-step — and the
    controller's counters must end where the full run leaves them.  A call whose two prompts differ must not be deduplicated."""
    dtype = torch.float16
    cfg, ocfg, sd, unet = setup(dev, dtype)
    L = cfg.sample_size
    S, gs, fstart = 8, 7.5, 3
    H = W = L * 8
    names = ou.lora_target_names(ocfg)
    bank = LoraBank(unet, [LoraAdapter(nm, {k: (a.to(dev), b.to(dev)) for k, (a, b) in ou.make_lora(ocfg, names, 8, 100 + c, 0.8, dtype)[0].items()})
                           for c, nm in enumerate(["c0", "c1", "style"])])
    concept = ConceptModels(unet, bank)
    pctl = pc.AttentionReplace([P, P], S, {"default_": 1.0}, 0.5, L // 4, L // 4, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    m1 = torch.zeros(H, W); m1[H // 4:, : W // 2] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 16:] = 1

    def request(seed, differ=False):
        pe1, pp1 = embeds(cfg, 1, seed, dtype); ne1, np1 = embeds(cfg, 1, seed + 50, dtype)
        pe = pe1.repeat(2, 1, 1)
        if differ:
            pe = pe.clone(); pe[1, 3] += 0.25
        regions = []
        for c in range(2):
            re_, rp_ = embeds(cfg, 2, seed + 10 + c, dtype)
            regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
        return dict(prompt_embeds=pe, negative_prompt_embeds=ne1.repeat(2, 1, 1), pooled_prompt_embeds=pp1.repeat(2, 1),
                    negative_pooled_prompt_embeds=np1.repeat(2, 1), region_prompt_embeds=regions, region_masks=[m1, m2],
                    latents=torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(seed)))

    def run(reqs, stage, dedup, use_graph):
        pctl.reset()
        traj = []
        pipe.generate_many(reqs, height=H, width=W, num_inference_steps=S, guidance_scale=gs, cross_attention_kwargs={"scale": 0.8},
                           controller=pctl, concept_models=concept, stage=stage, lora_list=["c0", "c1"], styleL=styleL, trajectory=traj,
                           fusion_start=fstart, use_graph=use_graph, dedup=dedup)
        assert (pctl.cur_step, pctl.cur_att_layer) == (S, 0)
        return torch.stack([t.cpu() for t in traj])

    one, two = [request(1)], [request(1), request(2)]
    for stage in (2, 1):
        full = run(one, stage, False, False)
        assert torch.equal(run(one, stage, True, False), full), f"stage {stage}: eager dedup differs"
        assert torch.equal(run(one, stage, True, True), full), f"stage {stage}: dedup through graphs differs (capture)"
        assert torch.equal(run(one, stage, True, True), full), f"stage {stage}: dedup through graphs differs (replay)"
        full2 = run(two, stage, False, False)
        assert torch.equal(full2[:, 0:1], full)
        assert torch.equal(run(two, stage, True, True), full2), f"stage {stage}: two requests"
        if stage == 2:      # the samples do part ways after the first fused step, and stay together before it
            assert torch.equal(full[fstart, 0, 0], full[fstart, 0, 1]) and not torch.equal(full[-1, 0, 0], full[-1, 0, 1])
    # different prompts for the two samples: dedup must fall back to the full batch by itself
    diff = [request(3, differ=True)]
    assert torch.equal(run(diff, 2, True, False), run(diff, 2, False, False))
    d = run(diff, 2, False, False)
    assert not torch.equal(d[0, 0, 0], d[0, 0, 1])


@pytest.mark.parametrize("lora_mode", ["merged", "segment"])
def test_style_lora_runs_on_the_main_pass_too(dev, lora_mode):
    """styleL (inference_lora.py:162-164, :253-254): the style LoRA is loaded into the MAIN pipe as well, so every main forward
    carries adapter 'style' at PEFT weight 1.0 x scale 0.8, while concept c runs set_adapters([c, 'style'], [0.7, 0.5])
    (lora_pipeline.py:588-591).  The style adapter here targets no to_q (partial coverage of an attention's projections)."""
    dtype = torch.float16
    cfg, ocfg, sd, unet = setup(dev, dtype)
    L = cfg.sample_size
    S, gs, fstart = 8, 7.5, 3
    H = W = L * 8
    neg_e, neg_p = embeds(cfg, 1, 1, dtype)
    pos_e, pos_p = embeds(cfg, 1, 2, dtype)
    pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    regions = []
    for c in range(2):
        re_, rp_ = embeds(cfg, 2, 10 + c, dtype)
        regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
    m1 = torch.zeros(H, W); m1[H // 4:, W // 16: W // 2] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 16: W - 8] = 1
    masks = [m1, m2]
    lat0 = torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(14))
    tid = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32)
    names = ou.lora_target_names(ocfg)
    style_names = [k for k in names if not k.endswith(".to_q")]
    ow, ofn = {}, {}
    for nm_, seed, tn in (("c0", 100, names), ("c1", 101, names), ("style", 102, style_names)):
        ow[nm_], ofn[nm_] = ou.make_lora(ocfg, tn, rank=8, seed=seed, scale=0.8, dtype=dtype)
    bank = LoraBank(unet, [LoraAdapter(k, {n: (a.to(dev), b.to(dev)) for n, (a, b) in w.items()}) for k, w in ow.items()])
    concept = ConceptModels(unet, bank)
    args = ([P, P], S, {"default_": 1.0}, 0.4, L // 4, L // 4)
    pctl = pc.AttentionReplace(*args, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    osch = osched.make("ddim", S)

    def oracle_run(stage, style):
        octl = oc.AttentionReplaceOracle(*args)
        octl.num_att_layers = pctl.num_att_layers
        attn = oc.reference_attn_fn(octl)
        ctx4 = torch.cat([ne, pe]); te4 = torch.cat([npp, pp])
        main_lora = ofn["style"] if style else None
        def main(x, i):
            return ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx4, te4, tid.repeat(4, 1), attn_fn=attn, lora=main_lora)
        def conc(c):
            ctx2 = torch.cat([regions[c][0], regions[c][1]]); te2 = torch.cat([regions[c][2], regions[c][3]])
            fn = (lambda key, x: 0.7 * ofn[f"c{c}"](key, x) + 0.5 * ofn["style"](key, x)) if style else ofn[f"c{c}"]
            return lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx2, te2, tid.repeat(2, 1), lora=fn)
        return opipe.denoise(main, [conc(c) for c in range(2)], osch, lat0 * osch.init_noise_sigma, S, gs, stage, masks=masks, fusion_start=fstart)

    outs = {}
    for stage in (1, 2):
        pctl.reset()
        out = pipe(output_type="latent", prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp,
                   height=H, width=W, num_inference_steps=S, guidance_scale=gs, latents=lat0, cross_attention_kwargs={"scale": 0.8},
                   controller=pctl, concept_models=concept, stage=stage, region_masks=masks, lora_list=["c0", "c1"], styleL=True,
                   region_prompt_embeds=regions, fusion_start=fstart, lora_mode=lora_mode).images.float().cpu()
        ref = oracle_run(stage, True)
        rms = ref.pow(2).mean().sqrt().item()
        rel = (out - ref).abs().max().item() / rms
        plain = oracle_run(stage, False)
        print(f"styleL {lora_mode} stage {stage}: rel err {rel:.2e}; style moves the oracle by {(ref - plain).abs().max().item() / rms:.2e} rms")
        assert rel < 2e-2, rel
        assert (ref - plain).abs().max().item() / rms > 10 * rel, "the style adapter must matter far more than the tolerance"
        outs[stage] = out


def test_stale_graphs_are_dropped_when_the_bank_is_rebuilt(dev):
    """call A (two masks, graphs captured) -> call B (one mask: the LoRA bank is rebuilt, weight stacks and cached
    K/V are re-allocated) -> call A again must not replay graphs that point at freed memory: equal to the eager result."""
    dtype = torch.float16
    cfg, ocfg, sd, unet = setup(dev, dtype)
    L = cfg.sample_size
    S, gs, fstart = 8, 7.5, 2
    H = W = L * 8
    names = ou.lora_target_names(ocfg)
    bank = LoraBank(unet, [LoraAdapter(nm, {k: (a.to(dev), b.to(dev)) for k, (a, b) in ou.make_lora(ocfg, names, 8, 100 + c, 0.8, dtype)[0].items()})
                           for c, nm in enumerate(["c0", "c1", "style"])])
    concept = ConceptModels(unet, bank)
    pctl = pc.AttentionReplace([P, P], S, {"default_": 1.0}, 0.5, L // 4, L // 4, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    m1 = torch.zeros(H, W); m1[H // 4:, : W // 2] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 16:] = 1

    def run(masks, use_graph, scale=0.8, styleL=False):
        pe1, pp1 = embeds(cfg, 1, 3, dtype); ne1, np1 = embeds(cfg, 1, 53, dtype)
        regions = []
        for c in range(2):
            re_, rp_ = embeds(cfg, 2, 13 + c, dtype)
            regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
        pctl.reset()
        return pipe(output_type="latent", prompt_embeds=pe1.repeat(2, 1, 1), negative_prompt_embeds=ne1.repeat(2, 1, 1), pooled_prompt_embeds=pp1.repeat(2, 1),
                    negative_pooled_prompt_embeds=np1.repeat(2, 1), height=H, width=W, num_inference_steps=S, guidance_scale=gs,
                    latents=torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(5)), cross_attention_kwargs={"scale": scale},
                    controller=pctl, concept_models=concept, stage=2, region_masks=masks, lora_list=["c0", "c1"], styleL=styleL,
                    region_prompt_embeds=regions, fusion_start=fstart, use_graph=use_graph).images.cpu()

    eager_a, eager_b = run([m1, m2], False), run([m1, None], False)
    ga1 = run([m1, m2], True)
    gb = run([m1, None], True)           # rebuilds the bank with ONE slot
    ga2 = run([m1, m2], True)            # rebuilds it with two: the first call's graphs are stale now
    assert torch.equal(ga1, eager_a) and torch.equal(gb, eager_b)
    assert torch.equal(ga2, eager_a), (ga2 - eager_a).abs().max()
    # the concept passes always run at the reference's hard-coded LoRA scale 0.8 (lora_pipeline.py:596): the caller's scale does not reach them
    assert torch.equal(run([m1, m2], False, scale=0.5), eager_a)
    # two engines alive at once (the reference's own sequence: stage 1, then stage 2 with the masks), one of them re-capturing after
    # the other's call rebuilt the bank: no captured graph may hold a pointer into ANOTHER engine's graph pool (round 3: the
    # GroupNorm scratch buffer did, and the full-size stage 1 -> stage 2 sequence faulted when the first engine dropped its graphs)
    def run_stage1(use_graph):
        pe1, pp1 = embeds(cfg, 1, 3, dtype); ne1, np1 = embeds(cfg, 1, 53, dtype)
        regions = []
        for c in range(2):
            re_, rp_ = embeds(cfg, 2, 13 + c, dtype)
            regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
        pctl.reset()
        return pipe(output_type="latent", prompt_embeds=pe1.repeat(2, 1, 1), negative_prompt_embeds=ne1.repeat(2, 1, 1), pooled_prompt_embeds=pp1.repeat(2, 1),
                    negative_pooled_prompt_embeds=np1.repeat(2, 1), height=H, width=W, num_inference_steps=S, guidance_scale=gs,
                    latents=torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(5)), cross_attention_kwargs={"scale": 0.8},
                    controller=pctl, concept_models=concept, stage=1, lora_list=["c0", "c1"], styleL=False,
                    region_prompt_embeds=regions, use_graph=use_graph).images.cpu()
    eager_1 = run_stage1(False)
    for _ in range(2):
        assert torch.equal(run_stage1(True), eager_1)
        assert torch.equal(run([m1, None], True), eager_b)      # rebuilds the bank: the stage-1 engine's graphs go stale ...
        assert torch.equal(run([m1, m2], True), eager_a)         # ... and this engine's, while the other two stay alive
    from omg_amd import ops as _ops
    assert not hasattr(_ops, "_gn_ws"), "no module-level device scratch: under capture it would live in one engine's graph pool"
    # ... but it is the scale of the main pass's style adapter: same engine key, the bank rebuilt with another style scale
    gs8 = run([m1, m2], True, scale=0.8, styleL=True)
    gs5 = run([m1, m2], True, scale=0.5, styleL=True)
    assert torch.equal(gs8, run([m1, m2], False, scale=0.8, styleL=True))
    assert torch.equal(gs5, run([m1, m2], False, scale=0.5, styleL=True))
    assert not torch.equal(gs5, gs8) and not torch.equal(gs8, eager_a)


@pytest.mark.parametrize("dtype", [torch.float16, pytest.param(torch.bfloat16, marks=pytest.mark.slow)])
def test_fifty_step_trajectory_error_growth(dev, dtype):
    """SURVEY §4.4 item 4 / VERDICT r1 next 1(c): the reference's own constants — 50 DDIM steps, fusion for i > 15, self-replace
    window of 20 steps (0.4 x 50), guidance 7.5, LoRA scale 0.8, two overlapping masks — on the tiny SDXL-topology UNet, fp16 AND
    bf16, against the fp32 oracle loop.  The per-step max |d| curve is written to gpurun_out/ (committed under profiles/)."""
    import json, os
    cfg, ocfg, sd, unet = setup(dev, dtype)
    L = cfg.sample_size
    S, gs, fstart = 50, 7.5, 15
    H = W = L * 8
    neg_e, neg_p = embeds(cfg, 1, 1, dtype)
    pos_e, pos_p = embeds(cfg, 1, 2, dtype)
    pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    regions = []
    for c in range(2):
        re_, rp_ = embeds(cfg, 2, 10 + c, dtype)
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
    bank = LoraBank(unet, [LoraAdapter(f"c{c}", {k: (a.to(dev), b.to(dev)) for k, (a, b) in ow[c].items()}) for c in range(2)])
    concept = ConceptModels(unet, bank)
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
    ref = opipe.denoise(main, [conc(0), conc(1)], osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=masks, fusion_start=fstart, record=rec)
    pctl.reset()
    traj = []
    pipe(output_type="latent", prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, height=H, width=W,
         num_inference_steps=S, guidance_scale=gs, latents=lat0, cross_attention_kwargs={"scale": 0.8}, controller=pctl,
         concept_models=concept, stage=2, region_masks=masks, lora_list=["c0", "c1"], styleL=False, region_prompt_embeds=regions,
         trajectory=traj, fusion_start=fstart)
    assert (pctl.cur_step, pctl.cur_att_layer) == (S, 0)
    errs = [(a.float().cpu() - b).abs().max().item() for a, b in zip(traj, rec)]
    rms = [b.pow(2).mean().sqrt().item() for b in rec]
    rel = [e / r for e, r in zip(errs, rms)]
    name = "fp16" if dtype == torch.float16 else "bf16"
    print(f"50-step stage-2 trajectory {name}: max|d|/rms at steps 1,10,16,17,20,30,40,50 = " +
          " ".join(f"{rel[i]:.2e}" for i in (0, 9, 15, 16, 19, 29, 39, 49)))
    # round 4: the same 50 steps by the oracle in the reference's own storage precision (every op's output rounded, oracle/precision.py):
    # how far the REFERENCE's arithmetic drifts from fp32 truth over the loop, and how far the HIP path is from it
    # (fp16 — the reference's dtype — only: the emulated loop is another ~95 s of host time per dtype; the bf16 curve of round 4 is committed
    # as profiles/r04_error_growth_bf16.json: bf16 oracle vs fp32 oracle 7.2e-2, HIP vs bf16 oracle 8.7e-2 at the worst step)
    # Round 6: the emulated twin (another ~95 s of host oracle) runs under OMG_RUN_SLOW=1 only — its curve is committed (profiles/r04_error_growth_fp16.json,
    # r05_..., and at full width r06_config0_fullwidth_loop.json); the driver's `-m gpu` keeps the fp32 comparison and its bound
    from oracle import precision as oprec
    rel_o = rel_h = None
    if dtype == torch.float16 and os.environ.get("OMG_RUN_SLOW") == "1":
        octl.reset()
        rec16 = []
        with oprec.rounding(dtype):
            opipe.denoise(main, [conc(0), conc(1)], osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=masks, fusion_start=fstart, record=rec16)
        rel_o = [(a - b).abs().max().item() / r for a, b, r in zip(rec16, rec, rms)]
        rel_h = [(a.float().cpu() - b).abs().max().item() / r for a, b, r in zip(traj, rec16, rms)]
        print(f"    {name} oracle vs fp32 oracle, worst step {max(rel_o):.2e} (last {rel_o[-1]:.2e});  HIP vs {name} oracle, worst {max(rel_h):.2e} (last {rel_h[-1]:.2e})")
        # the HIP path must not be further from exact arithmetic than twice the reference's own storage-precision arithmetic is
        assert max(rel) < 2.0 * max(rel_o) + 2e-3, (max(rel), max(rel_o))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"r04_error_growth_{name}.json"), "w") as f:
            json.dump({"what": "per-step max|latent - oracle latent| / oracle latent rms of a 50-step stage-2 call (tiny SDXL-topology UNet, DDIM, gs 7.5, fusion i>15, "
                               "self-replace 20 steps, 2 LoRA concepts with overlapping masks): the HIP path vs the fp32 CPU oracle loop, the oracle in the "
                               "reference's storage precision (every op's output rounded, oracle/precision.py) vs the fp32 oracle, and the HIP path vs that",
                       "dtype": name, "max_abs": errs, "oracle_latent_rms": rms, "max_abs_over_rms": rel,
                       "emulated_oracle_vs_fp32_oracle": rel_o, "hip_vs_emulated_oracle": rel_h}, f)
    except OSError:
        pass
    # measured on MI355X (profiles/r02_error_growth_*.json): the error grows over the first ~10 steps and then stays flat — fp16
    # 6.4e-3 of the latent rms at its worst step, bf16 4.6e-2; bound = measured + ~2x margin
    bound = 1.5e-2 if dtype == torch.float16 else 1e-1
    assert max(rel) < bound, (max(rel), rel)


@pytest.mark.parametrize("use_graph", [False, True])
def test_stage_two_resumes_from_the_stage_one_call_of_the_same_image(dev, use_graph):
    """SURVEY §7.4 / VERDICT r3 next 5: the reference runs stage 1 and stage 2 of an image with the SAME seed, prompts and kwargs
    (inference_lora.py:262-297) and fuses only for i > 15 (lora_pipeline.py:568), so the stage-2 call's steps 0..15 repeat the stage-1
    call.  With one :class:`StageCache` behind both calls the second starts at the first fused step from the stored latents: every
    later step's latents — both samples, every request — equal the uncached stage-2 call bit for bit, eager and through graphs, with
    ``dedup`` on either side; the controller's counters end where the full call leaves them; another seed, another prompt, another
    guidance scale or a re-configured scheduler miss and run in full."""
    from omg_amd.pipeline import StageCache
    dtype = torch.float16
    cfg, ocfg, sd, unet = setup(dev, dtype)
    L = cfg.sample_size
    S, gs, fstart = 9, 7.5, 3
    H = W = L * 8
    names = ou.lora_target_names(ocfg)
    bank = LoraBank(unet, [LoraAdapter(nm, {k: (a.to(dev), b.to(dev)) for k, (a, b) in ou.make_lora(ocfg, names, 8, 100 + c, 0.8, dtype)[0].items()})
                           for c, nm in enumerate(["c0", "c1"])])
    concept = ConceptModels(unet, bank)
    pctl = pc.AttentionReplace([P, P], S, {"default_": 1.0}, 0.5, L // 4, L // 4, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    m1 = torch.zeros(H, W); m1[H // 4:, : W // 2] = 1
    m2 = torch.zeros(H, W); m2[H // 4:, W // 2 - 16:] = 1

    def request(seed, eseed=None):
        eseed = seed if eseed is None else eseed
        pe1, pp1 = embeds(cfg, 1, eseed, dtype); ne1, np1 = embeds(cfg, 1, eseed + 50, dtype)
        regions = []
        for c in range(2):
            re_, rp_ = embeds(cfg, 2, eseed + 10 + c, dtype)
            regions.append((re_[0:1], re_[1:2], rp_[0:1], rp_[1:2]))
        return dict(prompt_embeds=pe1.repeat(2, 1, 1), negative_prompt_embeds=ne1.repeat(2, 1, 1), pooled_prompt_embeds=pp1.repeat(2, 1),
                    negative_pooled_prompt_embeds=np1.repeat(2, 1), region_prompt_embeds=regions, region_masks=[m1, m2],
                    latents=torch.randn(1, 4, L, L, generator=torch.Generator().manual_seed(seed)))

    def run(reqs, stage, cache=None, dedup=False, guidance=gs, drop_unc0=False):
        pctl.reset()
        traj = []
        out = pipe.generate_many(reqs, height=H, width=W, num_inference_steps=S, guidance_scale=guidance, cross_attention_kwargs={"scale": 0.8},
                                 controller=pctl, concept_models=concept, stage=stage, lora_list=["c0", "c1"], styleL=False, trajectory=traj,
                                 fusion_start=fstart, use_graph=use_graph, dedup=dedup, stage_cache=cache, drop_unc0=drop_unc0)
        assert (pctl.cur_step, pctl.cur_att_layer) == (S, 0)
        return out.cpu(), torch.stack([t.cpu() for t in traj])

    two = [request(1), request(2)]
    full, full_traj = run(two, 2)
    assert len(full_traj) == S
    cache = StageCache()
    s1, s1_traj = run(two, 1, cache, dedup=True)                     # stage 1 of the same images fills the cache ...
    assert len(cache.entries) == 2 and cache.misses == 2 and cache.hits == 0
    assert torch.equal(s1_traj[fstart], full_traj[fstart]), "steps 0..fusion_start of the two stages coincide"
    resumed, r_traj = run(two, 2, cache)                               # ... and stage 2 starts at the first fused step
    assert cache.hits == 2 and len(r_traj) == S - (fstart + 1)
    assert torch.equal(resumed, full) and torch.equal(r_traj, full_traj[fstart + 1:])
    assert torch.equal(resumed[:, 0], s1[:, 0]), "the base sample of stage 2 is the stage-1 image (SURVEY §7.4)"
    one_resumed, _ = run(two[1:], 2, cache, dedup=True)                # any subset of cached requests, any batching
    assert torch.equal(one_resumed, full[1:2])
    # ---- SURVEY 7.4's last item: the base sample is never fused, so its whole trajectory is the stage-1 call's; with it in the cache the
    # resumed steps run THREE main rows per request ([unc1, cond0, cond1]: `unc0` only ever fed the base sample's update) and still give
    # the same latents bit for bit — both samples, every step
    h0 = cache.hits
    dropped, d_traj = run(two, 2, cache, drop_unc0=True)
    assert cache.hits == h0 + 2 and torch.equal(dropped, full) and torch.equal(d_traj, full_traj[fstart + 1:])
    plans = [e.sh[True] for e in pipe._engines.values() if getattr(e, "sh", None) and True in e.sh]
    assert any(sh.n_main == 3 * len(two) and sh.rows == 3 * len(two) + 4 * len(two) for sh in plans), "three main rows + two concept pairs per request"
    assert torch.equal(run(two[:1], 2, cache, drop_unc0=True)[0], full[0:1])
    fresh = StageCache()                                               # a cache without the base trajectory: the plain resume, then the full call
    run(two, 2, fresh)                                                 # (a stage-2 call fills both the step-16 latents and the base trajectory)
    assert torch.equal(run(two, 2, fresh, drop_unc0=True)[0], full)
    # ---- misses: nothing is assumed
    for reqs, kw in (([request(3)], {}), ([request(1, eseed=7)], {}), ([request(1)], {"guidance": 5.0})):
        h0 = cache.hits
        got, t_ = run(reqs, 2, cache, **kw)
        assert cache.hits == h0 and len(t_) == S
        want, _ = run(reqs, 2, None, **kw)
        assert torch.equal(got, want)
    pipe.scheduler = make_scheduler("euler")                           # same step count, another table
    h0 = cache.hits
    run([request(1)], 2, cache)
    assert cache.hits == h0


def test_callbacks_and_negative_micro_conditioning_match_the_oracle_loop(dev):
    """Round 6 (VERDICT r5 missing 3): the reference kwargs that used to be refused.
    * `callback_on_step_end` (lora_pipeline.py:617-626) sees every step's latents and may replace them; `callback` (:629-632) every `callback_steps`;
    * `negative_original_size` / `negative_target_size` (:459-474): as the reference EXECUTES them the four main rows get [negative, positive,
      negative, positive] time ids (`cat([neg_ids, ids]).repeat(2, 1)`) — alternating, unlike the prompt embeddings' [neg, neg, pos, pos]."""
    dtype = torch.float16
    cfg, ocfg, sd, unet = setup(dev, dtype)
    L_ = cfg.sample_size
    S, gs, k_mod = 6, 7.5, 2
    H = W = L_ * 8
    neg_e, neg_p = embeds(cfg, 1, 1, dtype)
    pos_e, pos_p = embeds(cfg, 1, 2, dtype)
    pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
    lat0 = torch.randn(1, 4, L_, L_, generator=torch.Generator().manual_seed(14))
    args = ([P, P], S, {"default_": 1.0}, 0.4, L_ // 4, L_ // 4)
    pctl = pc.AttentionReplace(*args, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    neg_sizes = dict(negative_original_size=(H // 2, W // 2), negative_crops_coords_top_left=(8, 16), negative_target_size=(H, W // 2))

    def run(use_graph=False, **kw):
        pctl.reset()
        traj = []
        out = pipe(output_type="latent", prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp,
                   height=H, width=W, num_inference_steps=S, guidance_scale=gs, latents=lat0, controller=pctl, stage=1, trajectory=traj, use_graph=use_graph, **kw).images
        return out, torch.stack([t.cpu() for t in traj])

    # ---- oracle: the same loop with the reference's row order of the time ids and a scheduler whose step k is followed by `latents *= 0.5`
    osch = osched.make("ddim", S)

    class Halved:
        timesteps, init_noise_sigma = osch.timesteps, osch.init_noise_sigma
        scale_model_input = staticmethod(osch.scale_model_input)

        @staticmethod
        def step(eps, i, x):
            r = osch.step(eps, i, x)
            return r * 0.5 if i == k_mod else r

    def oracle(sched, tid4):
        octl = oc.AttentionReplaceOracle(*args)
        octl.num_att_layers = pctl.num_att_layers
        attn = oc.reference_attn_fn(octl)
        ctx4, te4 = torch.cat([ne, pe]), torch.cat([npp, pp])
        rec = []
        opipe.denoise(lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx4, te4, tid4, attn_fn=attn), [], sched, lat0 * osch.init_noise_sigma,
                      S, gs, 1, record=rec)
        return torch.stack(rec)

    pos_ids = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32)
    neg_ids = torch.tensor([[H // 2, W // 2, 8, 16, H, W // 2]], dtype=torch.float32)
    rel = lambda a, b: (a - b).abs().max().item() / b[-1].pow(2).mean().sqrt().item()

    # callbacks that change nothing: bitwise the plain call, every step seen once with the right index / timestep / latents
    _, plain = run()
    seen, legacy = [], []

    def on_end(p_, i, t, kw):
        assert p_ is pipe and set(kw) == {"latents", "prompt_embeds", "negative_prompt_embeds"}
        assert tuple(kw["prompt_embeds"].shape) == (4, 77, cfg.cross_attention_dim) and tuple(kw["negative_prompt_embeds"].shape) == (2, 77, cfg.cross_attention_dim)
        seen.append((i, float(t), kw["latents"].clone().cpu()))
        return {}

    _, with_cb = run(callback_on_step_end=on_end, callback_on_step_end_tensor_inputs=["latents", "prompt_embeds", "negative_prompt_embeds"],
                     callback=lambda i, t, l: legacy.append(i), callback_steps=2)
    assert torch.equal(with_cb, plain)
    assert [s_[0] for s_ in seen] == list(range(S)) and [s_[1] for s_ in seen] == [float(t) for t in osch.timesteps] and legacy == [0, 2, 4]
    assert all(torch.equal(s_[2], plain[i]) for i, s_ in enumerate(seen))
    # a callback that REPLACES the latents behind step k: the next step must start from them (and from the model input recomputed from them)
    halve = lambda p_, i, t, kw: {"latents": kw["latents"] * 0.5} if i == k_mod else {}
    _, got = run(callback_on_step_end=halve)
    ref = oracle(Halved, pos_ids.repeat(4, 1))
    assert torch.equal(got[:k_mod], plain[:k_mod]) and not torch.equal(got[k_mod], plain[k_mod])
    assert rel(got, ref) < 2e-2, rel(got, ref)
    _, got_g = run(use_graph=True, callback_on_step_end=halve)
    _, got_g2 = run(use_graph=True, callback_on_step_end=halve)
    assert torch.equal(got_g, got) and torch.equal(got_g2, got)
    with pytest.raises(Exception):
        run(callback_on_step_end=lambda p_, i, t, kw: {"prompt_embeds": kw["prompt_embeds"] * 2}, callback_on_step_end_tensor_inputs=["prompt_embeds"])
    # negative micro-conditioning, rows [neg, pos, neg, pos] as the reference builds them
    _, got_n = run(**neg_sizes)
    ref_n = oracle(osch, torch.cat([neg_ids, pos_ids, neg_ids, pos_ids]))
    ref_blocked = oracle(osch, torch.cat([neg_ids, neg_ids, pos_ids, pos_ids]))
    assert rel(got_n, ref_n) < 2e-2, rel(got_n, ref_n)
    assert rel(got_n, ref_blocked) > 5 * rel(got_n, ref_n), "the alternating row order of the reference, not the blocked one"
    assert not torch.equal(got_n, plain)
