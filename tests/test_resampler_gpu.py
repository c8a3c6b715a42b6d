"""InstantID's image_proj_model (IP-Adapter Resampler) on the HIP kernels vs oracle/resampler.py, which is pinned to the
reference's own Resampler class (tests/golden/resampler_golden.npz)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omg_amd.resampler import Resampler
from oracle import resampler as orr


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_resampler_matches_oracle(dev, dtype):
    cfg = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=64, output_dim=96, ff_mult=4)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in orr.param_shapes(*cfg.values()).items():
        if k == "latents":
            w = torch.randn(shp, generator=g) * cfg["dim"] ** -0.5
        elif k.endswith(".bias"):
            w = 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            w = torch.randn(shp, generator=g) * shp[-1] ** -0.5
        sd[k] = w.to(dtype)
    m = Resampler(**cfg, dtype=dtype, device=dev)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == orr.param_shapes(*cfg.values())
    m.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    x = torch.randn(3, 1, 64, generator=g).to(dtype)
    ref = orr.resampler_forward({k: v.float() for k, v in sd.items()}, x.float(), cfg["heads"])
    out = m(x.to(dev))
    assert out.shape == ref.shape == (3, 16, 96)
    err = (out.float().cpu() - ref).abs().max() / ref.pow(2).mean().sqrt()
    assert err < (2e-2 if dtype == torch.float16 else 1e-1), f"{err:.3e}"
