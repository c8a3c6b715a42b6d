"""ORACLE (test infrastructure): fp32 restatement of the IP-Adapter ``Resampler`` that InstantID uses as its
``image_proj_model`` (/root/reference src/ip_adapter/resampler.py:81-121; built with dim 1280, depth 4, dim_head 64, heads 20,
16 queries, embedding_dim 512, output_dim 2048 at src/pipelines/instantid_single_pieline.py:163-174; called on the
(zeros | face embedding) pair at :221-243).  It turns one 512-d face embedding into the 16 image-prompt tokens that the
IP-Adapter branch of the concept UNet attends to.

    latents = learned (1, Q, D) queries;  x = proj_in(x)
    per layer:  latents += to_out(softmax(q k^T / sqrt(dh)) v),  q = to_q(LN2(latents)),  k, v = to_kv(cat(LN1(x), LN2(latents)))
                latents += W2 gelu(W1 LN(latents))                                             (resampler.py:46-74, :9-16)
    return LN_out(proj_out(latents))

PARITY PINNED: src/ip_adapter/resampler.py imports cleanly here; tests/golden/resampler_golden.npz holds the output of the
reference's own ``Resampler`` on seeded weights (tests/golden/make_golden.py), and tests/test_oracle.py compares to 1e-5.
State-dict keys are the reference module's (``latents``, ``proj_in``, ``proj_out``, ``norm_out``, ``layers.i.0.{norm1,norm2,
to_q,to_kv,to_out}``, ``layers.i.1.{0,1,3}``).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def param_shapes(dim: int, depth: int, dim_head: int, heads: int, num_queries: int, embedding_dim: int, output_dim: int,
                 ff_mult: int = 4) -> Dict[str, tuple]:
    inner, ff = dim_head * heads, int(dim * ff_mult)
    out = {"latents": (1, num_queries, dim), "proj_in.weight": (dim, embedding_dim), "proj_in.bias": (dim,),
           "proj_out.weight": (output_dim, dim), "proj_out.bias": (output_dim,),
           "norm_out.weight": (output_dim,), "norm_out.bias": (output_dim,)}
    for i in range(depth):
        a, f = f"layers.{i}.0", f"layers.{i}.1"
        for n in ("norm1", "norm2"):
            out[f"{a}.{n}.weight"], out[f"{a}.{n}.bias"] = (dim,), (dim,)
        out[f"{a}.to_q.weight"], out[f"{a}.to_kv.weight"], out[f"{a}.to_out.weight"] = (inner, dim), (2 * inner, dim), (dim, inner)
        out[f"{f}.0.weight"], out[f"{f}.0.bias"] = (dim,), (dim,)
        out[f"{f}.1.weight"], out[f"{f}.3.weight"] = (ff, dim), (dim, ff)
    return out


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"].float(), sd[name + ".bias"].float(), 1e-5)


def resampler_forward(sd: Dict[str, Tensor], x: Tensor, heads: int) -> Tensor:
    """x: (B, n1, embedding_dim) -> (B, num_queries, output_dim)."""
    B = x.shape[0]
    lat = sd["latents"].float().repeat(B, 1, 1)
    x = F.linear(x.float(), sd["proj_in.weight"].float(), sd["proj_in.bias"].float())
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    for i in range(depth):
        a, f = f"layers.{i}.0", f"layers.{i}.1"
        xn, ln = _ln(sd, a + ".norm1", x), _ln(sd, a + ".norm2", lat)
        q = F.linear(ln, sd[a + ".to_q.weight"].float())
        k, v = F.linear(torch.cat([xn, ln], dim=1), sd[a + ".to_kv.weight"].float()).chunk(2, dim=-1)
        dh = q.shape[-1] // heads
        sp = lambda t: t.view(B, t.shape[1], heads, dh).transpose(1, 2)
        w = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * dh ** -0.5, dim=-1)
        o = (w @ sp(v)).transpose(1, 2).reshape(B, q.shape[1], heads * dh)
        lat = lat + F.linear(o, sd[a + ".to_out.weight"].float())
        h = F.gelu(F.linear(_ln(sd, f + ".0", lat), sd[f + ".1.weight"].float()))
        lat = lat + F.linear(h, sd[f + ".3.weight"].float())
    return _ln(sd, "norm_out", F.linear(lat, sd["proj_out.weight"].float(), sd["proj_out.bias"].float()))
