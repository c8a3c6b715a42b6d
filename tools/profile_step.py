"""Per-shape time breakdown of the GEMM/conv/attention launches of 2 plain + 2 fused denoising steps (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import controller as pc, ops
from omg_amd.pipeline import LoraMultiConceptPipeline, revise_regionally_controlnet_forward
from omg_amd.schedulers import make_scheduler
from omg_amd.synthetic import c2_inputs, c2_masks, make_concept_models
from omg_amd.unet import UNet2DConditionModel, UNetConfig
import contextlib, io

dev = torch.device("cuda:0"); dt = torch.float16
unet = UNet2DConditionModel(UNetConfig.sdxl(), dtype=dt, device=dev).init_synthetic_(0)
P = "a man and a woman"
ctl = pc.AttentionReplace([P, P], 50, {"default_": 1.0}, 0.4, 32, 32, device=dev, dtype=dt)
with contextlib.redirect_stdout(io.StringIO()):
    revise_regionally_controlnet_forward(unet, ctl)
concept = make_concept_models(unet, 2, 64)
pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
masks = c2_masks(1024, 1024, device=dev)
inp = c2_inputs(unet, 0)
kw = dict(height=1024, width=1024, guidance_scale=7.5, cross_attention_kwargs={"scale": 0.8}, controller=ctl, concept_models=concept,
          stage=2, region_masks=masks, lora_list=["concept0", "concept1"], styleL=False, output_type="latent", fusion_start=1, **inp)
extra = {}
if len(sys.argv) > 1:
    extra = eval(sys.argv[1])
pipe(num_inference_steps=4, **kw, **extra)   # warm
torch.cuda.synchronize()
t0 = time.perf_counter(); ctl.reset(); pipe(num_inference_steps=4, **kw, **extra); torch.cuda.synchronize(); wall = time.perf_counter() - t0
prof = ops.KernelProfiler(); ops.set_profiler(prof); ctl.reset()
pipe(num_inference_steps=4, **kw, **extra)
ops.set_profiler(None); torch.cuda.synchronize()
tot = 0
rows = prof.by_tag()
for (kind, tag), d in rows:
    tot += d["ms"]
print(f"wall (un-instrumented) {wall*1e3:.1f} ms for 4 steps; instrumented kernel sum {tot:.1f} ms")
for (kind, tag), d in rows[:40]:
    print(f"{kind:5s} {str(tag):48s} n={d['launches']:4d} ms={d['ms']:8.2f} ({100*d['ms']/tot:4.1f}%)  {d['flops']/d['ms']/1e9:7.1f} TF/s")
