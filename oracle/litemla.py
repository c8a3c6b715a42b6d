"""oracle/litemla.py — CPU restatement of EfficientViT's lightweight multi-scale linear attention, the segmentation hand-off
half of SURVEY §8(f) N4.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and the golden generator), never by omg_amd.

Restates /root/reference/src/efficientvit/models/nn/ops.py:
  * LiteMLA.__init__   :335-402   (qkv 1x1 conv; per scale a depthwise s x s conv followed by a 1x1 conv with 3*heads groups;
                                   proj 1x1 conv + BatchNorm2d)
  * relu_linear_att    :405-441   (fp32; q, k through ReLU; v padded with a column of ones; kv = k^T v; out = q kv;
                                   out[..., :-1] / (out[..., -1:] + eps))
  * forward            :443-455
  * ConvLayer          :37-78     (conv -> norm -> act; here norm in {None, bn2d in eval mode}, act None)
Parity pinned by tests/golden/litemla_golden.npz, produced by the reference's own LiteMLA class (tests/golden/make_golden.py).
State-dict keys are the reference's: qkv.conv.weight, aggreg.{i}.0.weight, aggreg.{i}.1.weight, proj.conv.weight,
proj.norm.{weight,bias,running_mean,running_var} (+ .bias entries when use_bias).
"""
from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def litemla_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, *, dim: int, scales: Sequence[int] = (5,), eps: float = 1.0e-15,
                    bn_eps: float = 1e-5) -> torch.Tensor:
    """x (B, Cin, H, W) fp32 -> (B, Cout, H, W).  heads = qkv rows / (3 dim)."""
    f = lambda k: sd[k].float()
    B, _, H, W = x.shape
    qkv = F.conv2d(x.float(), f("qkv.conv.weight"), sd.get("qkv.conv.bias"))                      # ops.py:445
    total3 = qkv.shape[1]
    heads = total3 // (3 * dim)
    ms = [qkv]
    for i, s in enumerate(scales):                                                                 # ops.py:447-448
        h = F.conv2d(qkv, f(f"aggreg.{i}.0.weight"), sd.get(f"aggreg.{i}.0.bias"), padding=s // 2, groups=total3)
        h = F.conv2d(h, f(f"aggreg.{i}.1.weight"), sd.get(f"aggreg.{i}.1.bias"), groups=3 * heads)
        ms.append(h)
    ms = torch.cat(ms, dim=1)                                                                      # ops.py:449
    # relu_linear_att, ops.py:405-441
    t = ms.reshape(B, -1, 3 * dim, H * W).transpose(-1, -2)                                        # (B, G, HW, 3 dim)
    q, k, v = t[..., :dim], t[..., dim:2 * dim], t[..., 2 * dim:]
    q, k = F.relu(q), F.relu(k)
    v = F.pad(v, (0, 1), mode="constant", value=1)
    kv = torch.matmul(k.transpose(-1, -2), v)                                                      # (B, G, dim, dim + 1)
    out = torch.matmul(q, kv)
    out = out[..., :-1] / (out[..., -1:] + eps)
    out = out.transpose(-1, -2).reshape(B, -1, H, W)
    # proj: conv -> BatchNorm2d (eval), ops.py:395-402, :452
    y = F.conv2d(out, f("proj.conv.weight"), sd.get("proj.conv.bias"))
    if "proj.norm.weight" in sd:
        y = F.batch_norm(y, f("proj.norm.running_mean"), f("proj.norm.running_var"), f("proj.norm.weight"), f("proj.norm.bias"), False, 0.0, bn_eps)
    return y


def init_state_dict(in_channels: int, out_channels: int, dim: int, scales: Sequence[int] = (5,), heads_ratio: float = 1.0, seed: int = 0,
                    dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded weights in the reference's key layout (shapes of LiteMLA.__init__, ops.py:353-402)."""
    g = torch.Generator().manual_seed(seed)
    heads = int(in_channels // dim * heads_ratio)
    T = heads * dim
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale)
    sd = {"qkv.conv.weight": r(3 * T, in_channels, 1, 1, scale=in_channels ** -0.5)}
    for i, s in enumerate(scales):
        sd[f"aggreg.{i}.0.weight"] = r(3 * T, 1, s, s, scale=1.0 / s)
        sd[f"aggreg.{i}.1.weight"] = r(3 * T, dim, 1, 1, scale=dim ** -0.5)
    Tm = T * (1 + len(scales))
    sd["proj.conv.weight"] = r(out_channels, Tm, 1, 1, scale=Tm ** -0.5)
    sd["proj.norm.weight"] = 1.0 + 0.2 * r(out_channels)
    sd["proj.norm.bias"] = 0.1 * r(out_channels)
    sd["proj.norm.running_mean"] = 0.1 * r(out_channels)
    sd["proj.norm.running_var"] = (1.0 + 0.3 * r(out_channels)).abs() + 0.1
    return {k: v.to(dtype) for k, v in sd.items()}
