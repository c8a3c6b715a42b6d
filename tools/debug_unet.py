"""Layer-by-layer comparison of the HIP UNet against the oracle (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd.unet import UNet2DConditionModel, UNetConfig, ResnetBlock2D, Transformer2DModel
from oracle import unet as ou

dt = torch.float16
dev = torch.device("cuda:0")
cfg, ocfg = UNetConfig.tiny(), ou.UNetConfig.tiny()
sd = ou.init_state_dict(ocfg, seed=0, dtype=dt)
unet = UNet2DConditionModel(cfg, dtype=dt, device=dev)
unet.load_state_dict({k: v.to(dt) for k, v in sd.items()})
g = torch.Generator().manual_seed(0)
L = cfg.sample_size
B = 2
x = torch.randn(B, 4, L, L, generator=g)
ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g).to(dt).float()
te = torch.randn(B, 64, generator=g).to(dt).float()
tid = torch.tensor([[L * 8.0, L * 8.0, 0, 0, L * 8.0, L * 8.0]] * B)
got = {}
def hook(name):
    def f(m, i, o):
        got[name] = o.float().cpu().permute(0, 3, 1, 2)
    return f
for name, m in unet.named_modules():
    if isinstance(m, (ResnetBlock2D, Transformer2DModel)):
        m.register_forward_hook(hook(name))
emb = unet.time_embed(981, B, te.to(dev).to(dt), tid.to(dev))
y = unet(x.to(dev), 981, encoder_hidden_states=ctx.to(dev).to(dt), added_cond_kwargs={"text_embeds": te.to(dev).to(dt), "time_ids": tid.to(dev)})[0]
taps = {}
ref = ou.unet_forward(sd, ocfg, x, 981, ctx, te, tid, taps=taps)
print("emb", (emb.float().cpu() - taps["emb"]).abs().max().item(), taps["emb"].abs().max().item())
for k, v in taps.items():
    if k in got:
        print(f"{k:40s} max|d|={(got[k]-v).abs().max().item():.3e}  ref_max={v.abs().max().item():.3f}")
print("out", (y.float().cpu() - ref).abs().max().item(), ref.abs().max().item())
