"""Per-layer-class sensitivity of the MX-fp8 mode, on the GPU box:

    python tools/mx8_sensitivity.py [--steps 50] [--out gpurun_out/r04/mx8_sensitivity.json]

Workload = tests/test_config4_gpu.py::test_fifty_step_error_growth_mx8's: BASELINE configs[1]'s loop (DDIM, fusion for i > 15, 20-step
self-replace window, guidance 7.5, two LoRA concepts with overlapping masks) on the SDXL topology at widths (128, 256, 512), against the
fp32 oracle loop.  For every layer class of omg_amd.unet.MX8_CLASSES: the final-latent error (rms and max over the oracle's latent rms)
with ONLY that class on the fp8 MFMA, and with every class BUT that one; then mixes (most harmful classes returned to fp16
one after the other).  The table decides omg_amd.unet.MX8_MIXED."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import controller as pc
from omg_amd.lora import LoraAdapter, LoraBank
from omg_amd.pipeline import ConceptModels, LoraMultiConceptPipeline, revise_regionally_controlnet_forward
from omg_amd.schedulers import make_scheduler
from omg_amd.unet import MX8_CLASSES, UNet2DConditionModel, UNetConfig
from oracle import controller as oc, pipeline as opipe, schedulers as osched, unet as ou

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--out", default="gpurun_out/r04/mx8_sensitivity.json")
a = ap.parse_args()
P = "a man and a woman walking on the street"
WIDE = dict(sample_size=16, block_out_channels=(128, 256, 512), transformer_layers_per_block=(1, 1, 2), attention_head_dim=(2, 4, 8),
            cross_attention_dim=128, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32)
dev, dtype = torch.device("cuda:0"), torch.float16
cfg, ocfg = UNetConfig(**WIDE), ou.UNetConfig(**WIDE)
sd = ou.init_state_dict(ocfg, seed=0, dtype=dtype)
unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev)
unet.load_state_dict({k: v.to(dtype) for k, v in sd.items()})


def emb(n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, 77, cfg.cross_attention_dim, generator=g).to(dtype).float(),
            torch.randn(n, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g).to(dtype).float())


L = cfg.sample_size
S, gs, fstart = a.steps, 7.5, 15
H = W = L * 8
pos_e, pos_p = emb(1, 2); neg_e, neg_p = emb(1, 1)
pe, ne, pp, npp = pos_e.repeat(2, 1, 1), neg_e.repeat(2, 1, 1), pos_p.repeat(2, 1), neg_p.repeat(2, 1)
regions = []
for c in range(2):
    re_, rp_ = emb(2, 10 + c)
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
concept = ConceptModels(unet, LoraBank(unet, [LoraAdapter(f"c{c}", {k: (x.to(dev), y.to(dev)) for k, (x, y) in ow[c].items()}) for c in range(2)]))
args = ([P, P], 50, {"default_": 1.0}, 0.4, L // 4, L // 4)
pctl = pc.AttentionReplace(*args, device=dev)
revise_regionally_controlnet_forward(unet, pctl)
pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
osch = osched.make("ddim", S)
octl = oc.AttentionReplaceOracle(*args)
octl.num_att_layers = pctl.num_att_layers
attn = oc.reference_attn_fn(octl)
ctx4 = torch.cat([ne, pe]); te4 = torch.cat([npp, pp])
main = lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx4, te4, tid.repeat(4, 1), attn_fn=attn)


def conc(c):
    ctx2 = torch.cat([regions[c][0], regions[c][1]]); te2 = torch.cat([regions[c][2], regions[c][3]])
    return lambda x, i: ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx2, te2, tid.repeat(2, 1), lora=olora[c])


t0 = time.time()
ref = opipe.denoise(main, [conc(0), conc(1)], osch, lat0 * osch.init_noise_sigma, S, gs, 2, masks=masks, fusion_start=fstart)
rms = ref.pow(2).mean().sqrt().item()
print(f"oracle loop: {time.time() - t0:.0f} s, final latent rms {rms:.3f}", flush=True)


def run(classes):
    n_on = unet.set_precision_classes(classes)
    pctl.reset()
    out = pipe(output_type="latent", prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, height=H, width=W,
               num_inference_steps=S, guidance_scale=gs, latents=lat0, cross_attention_kwargs={"scale": 0.8}, controller=pctl, concept_models=concept, stage=2,
               region_masks=masks, lora_list=["c0", "c1"], styleL=False, region_prompt_embeds=regions, fusion_start=fstart, use_graph=True).images
    d = out.float().cpu() - ref
    return {"classes": sorted(classes), "fp8_layers": n_on, "rms": d.pow(2).mean().sqrt().item() / rms, "max": d.abs().max().item() / rms}


rows = {"fp16": run([]), "all": run(MX8_CLASSES)}
for k in ("fp16", "all"):
    print(f"{k:22s} layers {rows[k]['fp8_layers']:3d}  rms {rows[k]['rms']:.3e}  max {rows[k]['max']:.3e}", flush=True)
alone, without = {}, {}
for c in MX8_CLASSES:
    alone[c] = run([c])
    without[c] = run([x for x in MX8_CLASSES if x != c])
    print(f"{c:10s} alone: layers {alone[c]['fp8_layers']:3d} rms {alone[c]['rms']:.3e} max {alone[c]['max']:.3e}   all but it: rms {without[c]['rms']:.3e} max {without[c]['max']:.3e}", flush=True)
# mixes: the classes returned to fp16 one after the other, most harmful first (harm = the error of the class alone in fp8)
mixes, cur = [], list(MX8_CLASSES)
for worst in sorted(MX8_CLASSES, key=lambda c: -alone[c]["rms"])[:-1]:
    cur = [x for x in cur if x != worst]
    r = run(cur)
    r["returned_to_fp16"] = worst
    mixes.append(r)
    print(f"mix: -{worst:10s} -> {len(cur)} classes {cur}: rms {r['rms']:.3e} max {r['max']:.3e}", flush=True)
unet.set_precision_classes([])
os.makedirs(os.path.dirname(a.out), exist_ok=True)
with open(a.out, "w") as f:
    json.dump({"what": "final-latent error / oracle latent rms of a %d-step stage-2 call (SDXL topology at widths (128, 256, 512), DDIM, gs 7.5, fusion i>15, 2 LoRA concepts, "
                       "overlapping masks) vs the fp32 oracle loop, by which layer classes run on the MX-fp8 MFMA" % S,
               "reference": rows, "class_alone_in_fp8": alone, "all_but_class_in_fp8": without, "mixes_most_harmful_class_first": mixes}, f, indent=1)
print("wrote", a.out)
