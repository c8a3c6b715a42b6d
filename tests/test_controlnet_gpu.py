"""ControlNet parity (-m gpu): omg_amd.controlnet.ControlNetModel vs oracle/controlnet.py on the tiny topology, and
the UNet consuming its residuals (down_block_additional_residuals / mid_block_additional_residual)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omg_amd.controlnet import ControlNetModel
from omg_amd.unet import UNet2DConditionModel, UNetConfig
from oracle import controlnet as ocn
from oracle import unet as ou


def test_controlnet_and_unet_residual_path(dev):
    dtype = torch.float16
    cfg, ocfg = UNetConfig.tiny(), ou.UNetConfig.tiny()
    csd = ocn.init_state_dict(ocfg, seed=3, dtype=dtype)
    usd = ou.init_state_dict(ocfg, seed=0, dtype=dtype)
    cn = ControlNetModel(cfg, dtype=dtype, device=dev)
    cn.load_state_dict({k: v.to(dtype) for k, v in csd.items()})
    unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev)
    unet.load_state_dict({k: v.to(dtype) for k, v in usd.items()})
    g = torch.Generator().manual_seed(0)
    L, B = cfg.sample_size, 2
    x = torch.randn(B, 4, L, L, generator=g)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g).to(dtype).float()
    te = torch.randn(B, 64, generator=g).to(dtype).float()
    tid = torch.tensor([[L * 8.0, L * 8.0, 0, 0, L * 8.0, L * 8.0]] * B)
    cond = torch.rand(1, 3, L * 8, L * 8, generator=g).to(dtype).float()           # one conditioning image for the batch
    added = {"text_embeds": te.to(dev).to(dtype), "time_ids": tid.to(dev)}
    down, mid = cn(x.to(dev), 981, encoder_hidden_states=ctx.to(dev).to(dtype), controlnet_cond=cond.to(dev), conditioning_scale=0.8,
                   added_cond_kwargs=added, return_dict=False)
    rdown, rmid = ocn.controlnet_forward(csd, ocfg, x, 981, ctx, cond.repeat(B, 1, 1, 1), 0.8, te, tid)
    assert len(down) == len(rdown) == 9
    for i, (a, b) in enumerate(zip(down + [mid], rdown + [rmid])):
        assert tuple(a.shape) == tuple(b.shape)
        err = (a.float().cpu() - b).abs().max().item()
        assert err < 2e-2, f"residual {i}: {err}"
    # cached conditioning embedding: second call must not recompute and must be identical
    down2, _ = cn(x.to(dev), 981, encoder_hidden_states=ctx.to(dev).to(dtype), controlnet_cond=next(iter(cn._cond_cache.values()))[2], conditioning_scale=0.8,
                  added_cond_kwargs=added)
    assert torch.equal(down2[0], down[0])
    y = unet(x.to(dev), 981, encoder_hidden_states=ctx.to(dev).to(dtype), added_cond_kwargs=added,
             down_block_additional_residuals=down, mid_block_additional_residual=mid)[0].float().cpu()
    ref = ou.unet_forward(usd, ocfg, x, 981, ctx, te, tid, down_block_additional_residuals=rdown, mid_block_additional_residual=rmid)
    plain = ou.unet_forward(usd, ocfg, x, 981, ctx, te, tid)
    err = (y - ref).abs().max().item()
    print(f"unet+controlnet max|d|={err:.3e}; effect of the residuals {(ref - plain).abs().max().item():.3f}")
    assert err < 3e-2 and (ref - plain).abs().max() > 0.1
    # NCHW (non channels_last) residuals, as a foreign ControlNet would pass them, take the conversion path
    y2 = unet(x.to(dev), 981, encoder_hidden_states=ctx.to(dev).to(dtype), added_cond_kwargs=added,
              down_block_additional_residuals=[d.contiguous() for d in down], mid_block_additional_residual=mid.contiguous())[0].float().cpu()
    assert torch.equal(y, y2)
