"""The oracle's un-pinned restatements vs vectors recorded from diffusers 0.25.0 itself (CPU, -m "not gpu").

`tests/golden/diffusers_golden.npz` is produced by `tests/golden/make_golden_diffusers.py` on a machine where
`diffusers==0.25.0` (the reference's pin, requirements.txt:5) is installed — it cannot be produced in the build container (no
network, no wheel).  While the file is absent these tests are SKIPPED and DESIGN.md §3 keeps saying "parity unpinned" for the UNet,
ControlNet, VAE decoder and schedulers; once someone commits it they run everywhere and move those rows to "pinned".
The script loads the oracle's seeded state dicts into diffusers' modules with strict=True, so the file's existence already proves
the key layout and every tensor shape."""
import os

import numpy as np
import pytest
import torch

from oracle import controlnet as ocn
from oracle import schedulers as osched
from oracle import unet as ou
from oracle import vae as ov

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "diffusers_golden.npz")
needs_file = pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/diffusers_golden.npz not generated (needs diffusers==0.25.0: "
                                                                   "python tests/golden/make_golden_diffusers.py)")
TINY = dict(sample_size=16, block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2), attention_head_dim=(1, 2, 4),
            cross_attention_dim=128, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32)
TOL = dict(rtol=1e-4, atol=2e-5)      # fp32 both sides; different summation orders only


@pytest.fixture(scope="module")
def gold():
    return np.load(PATH)


def t(a):
    return torch.from_numpy(np.asarray(a))


@needs_file
def test_unet_forward_matches_diffusers(gold):
    ocfg = ou.UNetConfig(**TINY)
    sd = ou.init_state_dict(ocfg, seed=0)
    for ts in (981, 21):
        y = ou.unet_forward(sd, ocfg, t(gold["x"]), ts, t(gold["ctx"]), t(gold["te"]), t(gold["tid"]))
        np.testing.assert_allclose(y.numpy(), gold[f"unet_t{ts}"], **TOL)


@needs_file
def test_controlnet_and_residual_path_match_diffusers(gold):
    ocfg = ou.UNetConfig(**TINY)
    sd, csd = ou.init_state_dict(ocfg, seed=0), ocn.init_state_dict(ocfg, seed=3)
    down, mid = ocn.controlnet_forward(csd, ocfg, t(gold["x"]), 981, t(gold["ctx"]), t(gold["cond"]), 0.8, t(gold["te"]), t(gold["tid"]))
    assert len(down) == sum(1 for k in gold.files if k.startswith("cn_down"))
    for i, d in enumerate(down):
        np.testing.assert_allclose(d.numpy(), gold[f"cn_down{i}"], **TOL)
    np.testing.assert_allclose(mid.numpy(), gold["cn_mid"], **TOL)
    y = ou.unet_forward(sd, ocfg, t(gold["x"]), 981, t(gold["ctx"]), t(gold["te"]), t(gold["tid"]),
                        down_block_additional_residuals=down, mid_block_additional_residual=mid)
    np.testing.assert_allclose(y.numpy(), gold["unet_with_cn"], **TOL)


@needs_file
def test_vae_decode_matches_diffusers(gold):
    vcfg = ov.VaeConfig.tiny()
    vsd = ov.init_state_dict(vcfg, seed=5)
    img = ov.decode(vsd, vcfg, t(gold["vae_z"]) / vcfg.scaling_factor)
    np.testing.assert_allclose(img.numpy(), gold["vae_image"], **TOL)


@needs_file
@pytest.mark.parametrize("name", ["ddim", "euler"])
@pytest.mark.parametrize("n", [50, 30, 10])
def test_scheduler_tables_and_steps_match_diffusers(gold, name, n):
    s = osched.make(name, n)
    np.testing.assert_allclose(np.asarray(s.timesteps, dtype=np.float64), gold[f"{name}{n}_timesteps"], rtol=0, atol=1e-9)
    assert abs(float(s.init_noise_sigma) - float(gold[f"{name}{n}_init_noise_sigma"])) < 1e-6
    if name == "euler":
        np.testing.assert_allclose(s.sigmas, gold[f"{name}{n}_sigmas"], rtol=1e-6, atol=1e-7)
    lat = gold[f"{name}{n}_lat0"]
    eps = torch.randn(3, 2, 4, 8, 8, generator=torch.Generator().manual_seed(10), dtype=torch.float64).numpy()
    for i in range(3):
        np.testing.assert_allclose(s.scale_model_input(lat, i), gold[f"{name}{n}_scaled"][i], rtol=1e-6, atol=1e-7)
        lat = s.step(eps[i], i, lat)
        np.testing.assert_allclose(lat, gold[f"{name}{n}_walk"][i], rtol=1e-6, atol=1e-7)


def test_recipe_is_committed_and_self_consistent():
    """Runs always: the generator script exists, names the reference's diffusers pin, and writes exactly the file this module reads."""
    script = os.path.join(os.path.dirname(PATH), "make_golden_diffusers.py")
    src = open(script).read()
    assert "diffusers==0.25.0" in src and "diffusers_golden.npz" in src and "strict=True" in src
    import ast
    ast.parse(src)
