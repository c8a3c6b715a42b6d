"""ORACLE (test infrastructure): restatement of the InstantID concept UNet's cross-attention,
``IPAttnProcessor2_0.__call__`` (/root/reference src/ip_adapter/attention_processor.py:324-424)
and of the plain ``AttnProcessor2_0`` (:207-293).

    out = to_out( SDPA(q, K_text, V_text) + scale * SDPA(q, to_k_ip(ip), to_v_ip(ip)) )

where the last ``num_tokens`` rows of ``encoder_hidden_states`` are the image-prompt tokens
(:362-366).  The write-only ``attn_map`` (:402-403) is not reproduced (never read anywhere).

PINNED: tests/golden/ip_adapter_golden.npz holds inputs, weights and outputs produced by the
reference's own class in this container (tests/golden/make_golden.py); tests/test_oracle.py
compares this restatement against it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _sdpa(q, k, v, heads):
    B, N, C = q.shape
    d = C // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)

    p = torch.softmax(split(q) @ split(k).transpose(-1, -2) * d ** -0.5, dim=-1)
    return (p @ split(v)).permute(0, 2, 1, 3).reshape(B, N, C)


def ip_cross_attention(hidden, ctx, to_q, to_k, to_v, to_out_w, to_out_b, to_k_ip, to_v_ip, heads, scale, num_tokens, lora=None, name=""):
    """``lora(key, x) -> delta`` (optional): a PEFT adapter wraps the Attention's OWN Linear modules, so ``attn.to_q`` / ``to_k`` / ``to_v``
    (on the text tokens) / ``to_out[0]`` carry it when the processor calls them (attention_processor.py:372-417); the processor's
    ``to_k_ip`` / ``to_v_ip`` are not LoRA targets."""
    def lin(key, x, w, b=None):
        y = F.linear(x, w, b)
        return y if lora is None else y + lora(f"{name}.{key}", x)
    end = ctx.shape[1] - num_tokens
    text, ip = ctx[:, :end], ctx[:, end:]
    q = lin("to_q", hidden, to_q)
    o = _sdpa(q, lin("to_k", text, to_k), lin("to_v", text, to_v), heads)
    o = o + scale * _sdpa(q, F.linear(ip, to_k_ip), F.linear(ip, to_v_ip), heads)
    return lin("to_out.0", o, to_out_w, to_out_b)


def self_attention(hidden, to_q, to_k, to_v, to_out_w, to_out_b, heads):
    o = _sdpa(F.linear(hidden, to_q), F.linear(hidden, to_k), F.linear(hidden, to_v), heads)
    return F.linear(o, to_out_w, to_out_b)


def make_ip_attn_fn(ip_weights, scale: float, num_tokens: int, base_attn_fn=None):
    """attn_fn for oracle.unet.unet_forward implementing the concept UNet of InstantID: self-attention is plain SDPA,
    cross-attention is IPAttnProcessor2_0 (the last ``num_tokens`` context rows are image-prompt tokens).
    ``ip_weights[attn2 module name] = (to_k_ip [C, Cx], to_v_ip [C, Cx])``."""
    from . import unet as ou

    def fn(name, heads, q, k, v, is_cross):
        return (base_attn_fn or ou.plain_attention)(name, heads, q, k, v, is_cross)

    def cross(sd, name, heads, x, ctx, lora=None):
        wk, wv = ip_weights[name]
        return ip_cross_attention(x, ctx, sd[name + ".to_q.weight"], sd[name + ".to_k.weight"], sd[name + ".to_v.weight"],
                                  sd[name + ".to_out.0.weight"], sd[name + ".to_out.0.bias"], wk, wv, heads, scale, num_tokens, lora=lora, name=name)

    fn.cross_override = cross
    return fn
