"""Find the first module whose output differs between two identical samples of a batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd.unet import UNet2DConditionModel, UNetConfig
from omg_amd import modules, attention, unet as U

dt = torch.float16
dev = torch.device("cuda:0")
cfg = UNetConfig.tiny()
unet = UNet2DConditionModel(cfg, dtype=dt, device=dev).init_synthetic_(0)
g = torch.Generator().manual_seed(0)
L = cfg.sample_size
x1 = torch.randn(1, 4, L, L, generator=g)
c1 = torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
t1 = torch.randn(1, 64, generator=g)
x = x1.repeat(4, 1, 1, 1).to(dev); ctx = c1.repeat(4, 1, 1).to(dev).to(dt); te = t1.repeat(4, 1).to(dev).to(dt)
tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]] * 4, device=dev)
bad = []
def hook(name):
    def f(m, i, o):
        if torch.is_tensor(o) and o.shape[0] == 4:
            d = (o[0].float() - o[1].float()).abs().max().item()
            d3 = (o[0].float() - o[3].float()).abs().max().item()
            if d > 0 or d3 > 0:
                bad.append((name, type(m).__name__, d, d3))
    return f
for name, m in unet.named_modules():
    if name:
        m.register_forward_hook(hook(name))
y = unet(x, 981, encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": te, "time_ids": tid})[0]
print("final diff", (y[0] - y[1]).abs().max().item(), (y[0] - y[3]).abs().max().item())
for b in bad[:12]:
    print(b)

# ---- pipeline level
from omg_amd import controller as pc
from omg_amd.pipeline import LoraMultiConceptPipeline, revise_regionally_controlnet_forward
from omg_amd.schedulers import make_scheduler
from omg_amd import ops
P = "a man and a woman"
ctl = pc.AttentionReplace([P, P], 8, {"default_": 1.0}, 0.4, L // 4, L // 4, device=dev)
revise_regionally_controlnet_forward(unet, ctl)
bad.clear()
y = unet(x, 981, encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": te, "time_ids": tid})[0]
print("with controller: final diff", (y[0] - y[1]).abs().max().item(), (y[2] - y[3]).abs().max().item(), (y[0] - y[3]).abs().max().item())
for b in bad[:8]:
    print(b)
pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
ctl.reset()
traj = []
pe = c1.repeat(2, 1, 1); pp = t1.repeat(2, 1)
ne = torch.randn(1, 77, cfg.cross_attention_dim, generator=g).repeat(2, 1, 1); npp = torch.randn(1, 64, generator=g).repeat(2, 1)
out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, height=L * 8, width=L * 8,
           num_inference_steps=8, guidance_scale=7.5, latents=x1, controller=ctl, stage=1, lora_list=[], trajectory=traj).images
for i, t in enumerate(traj):
    print("step", i, "lat diff", (t[0] - t[1]).abs().max().item())
