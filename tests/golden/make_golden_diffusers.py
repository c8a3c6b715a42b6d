"""Pin the UNPINNED part of the oracle against the library the reference actually runs on.

    pip install diffusers==0.25.0        (requirements.txt:5 of the reference; not installable in the build container: no network)
    python tests/golden/make_golden_diffusers.py        -> tests/golden/diffusers_golden.npz

`oracle/{unet,controlnet,vae,schedulers}.py` restate un-vendored diffusers 0.25.0 from its published algorithm; nothing in
the build container can check them.  This script is the recipe for anyone WITH diffusers: it builds diffusers' own
`UNet2DConditionModel`, `ControlNetModel`, `AutoencoderKL` (decoder side) and `DDIMScheduler` / `EulerDiscreteScheduler` at the
SDXL topology in small widths, loads the SAME seeded weights the oracle uses (the state-dict key layout is shared — a strict
`load_state_dict` is the first check), runs them in fp32 on the CPU and records inputs and outputs.
`tests/test_oracle_diffusers.py` then compares the oracle with these vectors (and is skipped while the file is absent).

Everything is seeded and CPU fp32; the file is ~2 MB.  Only `diffusers`, `torch`, `numpy` and this repository are needed.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import controlnet as ocn          # noqa: E402
from oracle import unet as ou                 # noqa: E402
from oracle import vae as ov                  # noqa: E402

TINY = dict(sample_size=16, block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2), attention_head_dim=(1, 2, 4),
            cross_attention_dim=128, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32)
SDXL_SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
                  timestep_spacing="leading", prediction_type="epsilon")


def main():
    import diffusers
    from diffusers import AutoencoderKL, ControlNetModel, DDIMScheduler, EulerDiscreteScheduler, UNet2DConditionModel
    out = {"diffusers_version": np.array(diffusers.__version__)}
    torch.manual_seed(0)
    ocfg = ou.UNetConfig(**TINY)
    common = dict(in_channels=4, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                  block_out_channels=TINY["block_out_channels"], layers_per_block=2, transformer_layers_per_block=TINY["transformer_layers_per_block"],
                  attention_head_dim=TINY["attention_head_dim"], cross_attention_dim=TINY["cross_attention_dim"], use_linear_projection=True,
                  addition_embed_type="text_time", addition_time_embed_dim=TINY["addition_time_embed_dim"],
                  projection_class_embeddings_input_dim=TINY["projection_class_embeddings_input_dim"], norm_num_groups=32)
    g = torch.Generator().manual_seed(1)
    B, L = 2, TINY["sample_size"]
    x = torch.randn(B, 4, L, L, generator=g)
    ctx = torch.randn(B, 77, TINY["cross_attention_dim"], generator=g)
    te = torch.randn(B, 64, generator=g)
    tid = torch.tensor([[L * 8.0, L * 8.0, 0, 0, L * 8.0, L * 8.0]] * B)
    cond = torch.rand(B, 3, L * 8, L * 8, generator=g)
    out.update(x=x.numpy(), ctx=ctx.numpy(), te=te.numpy(), tid=tid.numpy(), cond=cond.numpy())
    added = {"text_embeds": te, "time_ids": tid}

    # ---- UNet2DConditionModel: strict key / shape equality, two timesteps, with and without ControlNet residuals
    unet = UNet2DConditionModel(sample_size=L, out_channels=4, up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), **common).eval()
    usd = ou.init_state_dict(ocfg, seed=0)
    missing = unet.load_state_dict(usd, strict=True)
    out["unet_keys_match"] = np.array(1)
    cn = ControlNetModel(conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256), **common).eval()
    csd = ocn.init_state_dict(ocfg, seed=3)
    cn.load_state_dict(csd, strict=True)
    with torch.no_grad():
        for t in (981, 21):
            out[f"unet_t{t}"] = unet(x, t, encoder_hidden_states=ctx, added_cond_kwargs=added, return_dict=False)[0].numpy()
        down, mid = cn(x, 981, encoder_hidden_states=ctx, controlnet_cond=cond, conditioning_scale=0.8, added_cond_kwargs=added, return_dict=False)
        for i, d in enumerate(down):
            out[f"cn_down{i}"] = d.numpy()
        out["cn_mid"] = mid.numpy()
        out["unet_with_cn"] = unet(x, 981, encoder_hidden_states=ctx, added_cond_kwargs=added, down_block_additional_residuals=down,
                                   mid_block_additional_residual=mid, return_dict=False)[0].numpy()

    # ---- AutoencoderKL.decode (the tail of the reference's call, lora_pipeline.py:635-661)
    vcfg = ov.VaeConfig.tiny()
    vae = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * len(vcfg.block_out_channels),
                        up_block_types=("UpDecoderBlock2D",) * len(vcfg.block_out_channels), block_out_channels=vcfg.block_out_channels,
                        layers_per_block=vcfg.layers_per_block, latent_channels=4, norm_num_groups=32, scaling_factor=vcfg.scaling_factor).eval()
    vsd = ov.init_state_dict(vcfg, seed=5)
    have = vae.state_dict()
    dec_keys = [k for k in have if k.startswith(("decoder.", "post_quant_conv."))]
    assert sorted(dec_keys) == sorted(vsd), (sorted(set(dec_keys) ^ set(vsd))[:8])
    vae.load_state_dict({**have, **vsd}, strict=True)
    z = torch.randn(1, 4, 8, 8, generator=g)
    with torch.no_grad():
        out["vae_z"] = z.numpy()
        out["vae_image"] = vae.decode(z / vae.config.scaling_factor, return_dict=False)[0].numpy()

    # ---- schedulers: SDXL-base's configuration; timesteps, sigmas, init_noise_sigma and a 3-step walk on fixed eps
    for name, cls in (("ddim", DDIMScheduler), ("euler", EulerDiscreteScheduler)):
        for n in (50, 30, 10):
            kw = dict(SDXL_SCHED)
            if name == "ddim":
                kw.update(clip_sample=False, set_alpha_to_one=False)
            s = cls(**kw)
            s.set_timesteps(n)
            out[f"{name}{n}_timesteps"] = s.timesteps.numpy().astype(np.float64)
            out[f"{name}{n}_init_noise_sigma"] = np.array(float(s.init_noise_sigma))
            if hasattr(s, "sigmas"):
                out[f"{name}{n}_sigmas"] = s.sigmas.numpy().astype(np.float64)
            lat = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(9), dtype=torch.float64) * float(s.init_noise_sigma)
            eps = torch.randn(3, 2, 4, 8, 8, generator=torch.Generator().manual_seed(10), dtype=torch.float64)
            out[f"{name}{n}_lat0"] = lat.numpy()
            walk, scaled = [], []
            for i, t in enumerate(s.timesteps[:3]):
                scaled.append(s.scale_model_input(lat, t).numpy())
                lat = s.step(eps[i], t, lat, return_dict=False)[0]
                walk.append(lat.numpy())
            out[f"{name}{n}_scaled"] = np.stack(scaled)
            out[f"{name}{n}_walk"] = np.stack(walk)
    dst = os.path.join(HERE, "diffusers_golden.npz")
    np.savez_compressed(dst, **out)
    print(dst, len(out), "arrays; diffusers", diffusers.__version__)


if __name__ == "__main__":
    main()
