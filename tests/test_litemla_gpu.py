"""EfficientViT LiteMLA on the HIP kernels (-m gpu): the two new kernels against torch fp32 on the same rounded inputs, the module
against the golden vectors of the reference's own class (tests/golden/litemla_golden.npz) and against oracle/litemla.py at
EfficientViT-SAM's real shape (1024^2 image -> 64 x 64 tokens, 512 channels, dim 32)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from omg_amd import ops
from omg_amd.litemla import LiteMLA
from oracle import litemla as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "litemla_golden.npz")
DTYPES = [torch.float16, torch.bfloat16]


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float16):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,k", [(2, 8, 8, 48, 5), (1, 13, 7, 192, 3), (3, 6, 10, 96, 5)])
def test_dwconv2d(dev, dtype, B, H, W, C, k):
    wide = rnd(B * H * W, C + 64, seed=1, dtype=dtype).to(dev)          # the kernel reads a column slice of a wider buffer
    x = wide[:, 32:32 + C]
    w = rnd(C, 1, k, k, seed=2, scale=1.0 / k, dtype=dtype)
    got = ops.dwconv2d(x, w.reshape(C, k * k).t().contiguous().to(dev), B, H, W, k)
    ref = F.conv2d(x.float().cpu().reshape(B, H, W, C).permute(0, 3, 1, 2), w.float(), padding=k // 2, groups=C).permute(0, 2, 3, 1).reshape(-1, C)
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1.6e-2, atol=1.6e-2)
    torch.testing.assert_close(got.float().cpu(), ref, **tol)


def _rla_ref(qkv, B, HW, G, dim, eps):
    t = qkv.float().reshape(B, HW, G, 3 * dim).permute(0, 2, 1, 3)                   # (B, G, HW, 3 dim): ops.py:409-419
    q, k, v = F.relu(t[..., :dim]), F.relu(t[..., dim:2 * dim]), t[..., 2 * dim:]
    v = F.pad(v, (0, 1), value=1.0)
    out = q @ (k.transpose(-1, -2) @ v)
    out = out[..., :-1] / (out[..., -1:] + eps)
    return out.permute(0, 2, 1, 3).reshape(B * HW, G * dim)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,G,dim", [(2, 64, 4, 16), (1, 300, 6, 32), (3, 129, 5, 8), (1, 4096, 32, 32)])
def test_relu_linear_att(dev, dtype, B, HW, G, dim):
    qkv = rnd(B * HW, G * 3 * dim, seed=3, dtype=dtype)
    got = ops.relu_linear_att(qkv.to(dev), B, HW, G, dim, 1e-15)
    ref = _rla_ref(qkv, B, HW, G, dim, 1e-15)
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1.6e-2, atol=1.6e-2)
    torch.testing.assert_close(got.float().cpu(), ref, **tol)
    # size-independent identity: permuting the tokens permutes the output rows and nothing else (kv is a sum over tokens)
    perm = torch.randperm(HW, generator=torch.Generator().manual_seed(4))
    q2 = qkv.reshape(B, HW, -1)[:, perm].reshape(B * HW, -1).contiguous()
    got2 = ops.relu_linear_att(q2.to(dev), B, HW, G, dim, 1e-15).float().cpu().reshape(B, HW, -1)
    torch.testing.assert_close(got2, got.float().cpu().reshape(B, HW, -1)[:, perm], rtol=tol["rtol"], atol=tol["atol"])


def _module(sd, cin, cout, dim, scales, dtype, dev):
    m = LiteMLA(cin, cout, dim=dim, scales=scales, dtype=dtype, device=dev)
    res = m.load_state_dict({k: v.to(dtype) if v.dtype.is_floating_point and "running" not in k else v for k, v in sd.items()}, strict=False)
    assert all(k.endswith("num_batches_tracked") for k in res.missing_keys) and not res.unexpected_keys, res
    return m


@pytest.mark.parametrize("dtype", DTYPES)
def test_module_matches_the_reference_class_golden(dev, dtype):
    g = np.load(GOLD)
    for tag in sorted({k[:-4] for k in g.files if k.endswith("_cfg")}):
        cin, cout, dim, B, H, W, *scales = g[f"{tag}_cfg"].tolist()
        sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_sd_")}
        m = _module(sd, cin, cout, dim, tuple(scales), dtype, dev)
        x, y = torch.from_numpy(g[f"{tag}_x"]), torch.from_numpy(g[f"{tag}_y"])
        got = m(x.to(dtype).to(dev)).float().cpu()
        assert got.shape == y.shape
        rel = (got - y).abs().max().item() / y.pow(2).mean().sqrt().item()
        print(f"LiteMLA {tag} {dtype}: max |d| / rms vs the reference class (fp32) {rel:.2e}")
        assert rel < (3e-2 if dtype == torch.float16 else 2e-1), (tag, rel)


@pytest.mark.parametrize("dtype", DTYPES)
def test_module_at_efficientvit_sam_shape_vs_oracle(dev, dtype):
    """64 x 64 tokens (a 1024^2 image at stride 16), 512 channels, dim 32 (16 heads): the shape the segmenter runs between the
    stages; oracle on the SAME rounded weights and input, plus the ResidualBlock shortcut."""
    cin, dim, B, H, W = 512, 32, 2, 64, 64
    sd = ol.init_state_dict(cin, cin, dim, (5,), seed=7, dtype=dtype)
    m = _module(sd, cin, cin, dim, (5,), dtype, dev)
    x = rnd(B, cin, H, W, seed=8, dtype=dtype)
    ref = ol.litemla_forward({k: v.float() for k, v in sd.items()}, x.float(), dim=dim)
    got = m(x.to(dev)).float().cpu()
    rms = ref.pow(2).mean().sqrt().item()
    rel = (got - ref).abs().max().item() / rms
    print(f"LiteMLA 512ch 64x64 {dtype}: max |d| / rms vs oracle {rel:.2e}")
    assert rel < (2e-2 if dtype == torch.float16 else 1.5e-1)
    got_r = m(x.to(dev), residual=True).float().cpu()
    rel_r = (got_r - (ref + x.float())).abs().max().item() / rms
    assert rel_r < (2e-2 if dtype == torch.float16 else 1.5e-1)


def test_litemla_has_no_cpu_fallback():
    from omg_amd import _lib as L
    m = LiteMLA(64, 64, dim=16, dtype=torch.float16, device="cpu")
    with pytest.raises(L.OmgHipError):
        m(torch.zeros(1, 64, 4, 4, dtype=torch.float16))
