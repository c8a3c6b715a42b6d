"""Attention module + attention processors for the MI355X UNet.

Boundary B1: the diffusers attention-processor protocol
``proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
**cross_attention_kwargs) -> Tensor`` as used by ``RegionControlNet_AttnProcessor``
(/root/reference src/pipelines/lora_pipeline.py:61-133), installed with ``set_processor`` on
modules whose class is named ``Attention`` (:139-140).

Three processors are provided:

* :class:`FusedAttnProcessor` — plain softmax(QK^T)V (what xformers / SDPA compute for the
  concept UNet, src/ip_adapter/attention_processor.py:207-293).
* :class:`RegionControlNet_AttnProcessor` — same name and constructor as the reference's class.
  With an OMG ``AttentionReplace`` controller whose mapper is the identity and alpha is 1 (always,
  in OMG's flows) the controller's in-place edit ``probs[cond_i] := probs[cond_0]`` is executed as
  *probability borrowing* inside the flash-attention kernel (never materialising the
  ``(B*heads, N, N)`` tensor).  For any other controller it falls back to the reference's literal
  sequence (scores -> softmax -> controller(probs) -> bmm) on HIP kernels ("protocol mode").
* :class:`IPAttnProcessor2_0` — text + scale * image-prompt attention of the InstantID concept
  UNet (src/ip_adapter/attention_processor.py:296-424).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import _lib as L
from . import ops
from .modules import Dropout, Linear, LoraState, bump_pointer_epoch, pointer_epoch  # noqa: F401


class Attention(nn.Module):
    """The slice of ``diffusers.models.attention_processor.Attention`` that OMG touches."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int], heads: int, dim_head: int = 64,
                 dtype=torch.float16, device=None):
        super().__init__()
        if dim_head != 64:
            raise L.OmgHipError("the gfx950 attention kernel is specialised for head_dim 64 (all SDXL layers)")
        inner = heads * dim_head
        self.heads, self.scale, self.inner_dim = heads, dim_head ** -0.5, inner
        self.is_cross = cross_attention_dim is not None
        kv_dim = cross_attention_dim if self.is_cross else query_dim
        self.to_q = Linear(query_dim, inner, bias=False, dtype=dtype, device=device)
        self.to_k = Linear(kv_dim, inner, bias=False, dtype=dtype, device=device)
        self.to_v = Linear(kv_dim, inner, bias=False, dtype=dtype, device=device)
        self.to_out = nn.ModuleList([Linear(inner, query_dim, bias=True, dtype=dtype, device=device), Dropout()])
        # attributes the reference processor reads (lora_pipeline.py:81-131)
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = FusedAttnProcessor()
        self._qkv = None          # fused [3*inner, C] weight for self-attention
        self._kv = None           # fused [2*inner, Cx] weight for cross-attention
        self._qkv_slots = None    # merged-LoRA mode: [1+slots, 3*inner, C]
        self._kv_slots = None
        self._kv_cache = {}       # key -> (K, V^T, ctx) projections of constant encoder_hidden_states
        # IP-Adapter branch (omg_amd.ip_adapter.IPAdapter): fused [to_k_ip ; to_v_ip] weight, scale, token count
        self.ip_kv_weight: Optional[torch.Tensor] = None
        self.ip_scale, self.ip_tokens = 1.0, 16
        self._ip_cache = None
        self._mx8 = {}            # MX-fp8 images of the fused q|k|v weight (stack): "qkv", "qkv_slots"

    # ---- diffusers API
    def set_processor(self, processor) -> None:
        self.processor = processor

    def get_processor(self):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is not None:
            raise L.OmgHipError("attention masks are not used on OMG's path (always None) and are not supported")
        return None

    def head_to_batch_dim(self, t: torch.Tensor) -> torch.Tensor:
        B, N, _ = t.shape
        return t.reshape(B, N, self.heads, 64).permute(0, 2, 1, 3).reshape(B * self.heads, N, 64)

    def batch_to_head_dim(self, t: torch.Tensor) -> torch.Tensor:
        BH, N, d = t.shape
        return t.reshape(BH // self.heads, self.heads, N, d).permute(0, 2, 1, 3).reshape(BH // self.heads, N, self.heads * d)

    def get_attention_scores(self, query: torch.Tensor, key: torch.Tensor, attention_mask=None) -> torch.Tensor:
        """softmax(scale * Q K^T) as an explicit (B*heads, Nq, Nkv) tensor (protocol mode only)."""
        if attention_mask is not None:
            raise L.OmgHipError("attention masks are not supported")
        q, k = query.contiguous(), key.contiguous()
        return ops.attn_probs(q, k, 1, self.scale)      # batch' = B*heads, one head of 64

    def _apply(self, fn, *a, **k):        # .to() / .half(): the fused weight images and K/V caches belong to the old storage
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    # ---- fused projections
    def invalidate_packed(self):
        self._qkv = self._kv = self._qkv_slots = self._kv_slots = None
        self._kv_cache = {}
        self._mx8 = {}
        self._ip_cache = None
        bump_pointer_epoch()

    def qkv_weight(self) -> torch.Tensor:
        if self._qkv is None:
            self._qkv = torch.cat([self.to_q.weight.data, self.to_k.weight.data, self.to_v.weight.data], dim=0).contiguous()
        return self._qkv

    def kv_weight(self) -> torch.Tensor:
        if self._kv is None:
            self._kv = torch.cat([self.to_k.weight.data, self.to_v.weight.data], dim=0).contiguous()
        return self._kv

    def _has_lora(self) -> bool:
        """Segment mode with an adapter on ANY of to_q / to_k / to_v: the projections then run as separate GEMMs, each adding
        its own second K-segment (a projection without adapter weights simply has none)."""
        st = self.to_q.lora_state
        return st is not None and not st.merged and any(l.lora_down is not None for l in (self.to_q, self.to_k, self.to_v))

    def _merged(self) -> Optional[LoraState]:
        """Merged mode: LoraBank.build gives all three projections a ``w_slots`` stack as soon as one of them is targeted."""
        st = self.to_q.lora_state
        if st is None or not st.merged:
            return None
        have = [l.w_slots is not None for l in (self.to_q, self.to_k, self.to_v)]
        if any(have) and not all(have):
            raise L.OmgHipError("merged LoRA slots cover only part of to_q/to_k/to_v; rebuild the LoraBank")
        return st if all(have) else None

    def qkv_slots(self) -> torch.Tensor:
        if self._qkv_slots is None:
            self._qkv_slots = torch.cat([self.to_q.w_slots, self.to_k.w_slots, self.to_v.w_slots], dim=1).contiguous()
        return self._qkv_slots

    def kv_slots(self) -> torch.Tensor:
        if self._kv_slots is None:
            self._kv_slots = torch.cat([self.to_k.w_slots, self.to_v.w_slots], dim=1).contiguous()
        return self._kv_slots

    def project_self(self, x):
        """x (B,N,C) (a tensor, or the MX-fp8 output of the preceding LayerNorm) -> q, k views of one fused buffer, and V^T."""
        B, N, C = x.shape
        inner = self.inner_dim
        if isinstance(x, ops.Mx8Tensor) or (self.to_q.mx8 and C % 128 == 0 and not self._has_lora()):
            if self._has_lora():
                raise L.OmgHipError("MX-fp8 activations cannot feed a segment-mode LoRA projection")
            xq = x if isinstance(x, ops.Mx8Tensor) else ops.quant_mx8(x)
            st = self._merged()
            if st is not None:
                if "qkv_slots" not in self._mx8:
                    self._mx8["qkv_slots"] = ops.quant_mx8(self.qkv_slots().reshape(-1, C))
                qkv = ops.gemm_mx8(xq, self._mx8["qkv_slots"], out_dtype=self.to_q.weight.dtype, groups=st.groups,
                                   w_group_adapter=st.group_adapter, n_per_adapter=3 * inner)
            else:
                if "qkv" not in self._mx8:
                    self._mx8["qkv"] = ops.quant_mx8(self.qkv_weight())
                qkv = ops.gemm_mx8(xq, self._mx8["qkv"], out_dtype=self.to_q.weight.dtype)
            qkv = qkv.view(B, N, 3 * inner)
            return qkv[:, :, :inner], qkv[:, :, inner:2 * inner], ops.value_operand(qkv[:, :, 2 * inner:], self.heads)
        if self._has_lora():
            q = self.to_q(x); k = self.to_k(x); v = self.to_v(x)
        else:
            st = self._merged()
            if st is not None:
                qkv = ops.gemm(x.reshape(B * N, C), self.qkv_slots(), groups=st.groups, w_group_adapter=st.group_adapter)
            else:
                qkv = ops.gemm(x.reshape(B * N, C), self.qkv_weight())
            qkv = qkv.view(B, N, 3 * inner)
            q, k, v = qkv[:, :, :inner], qkv[:, :, inner:2 * inner], qkv[:, :, 2 * inner:]
        return q, k, ops.value_operand(v, self.heads)

    def project_cross(self, ctx: torch.Tensor):
        """ctx (B,Nk,Cx) -> K view and V^T of a constant ``encoder_hidden_states``.

        Cached per (ctx shape, LoRA state): a hit returns the stored projections; a miss for a shape seen before
        recomputes INTO the stored tensors (pointers stay valid for captured hipGraphs); entries keep ``ctx`` alive."""
        st = self.to_k.lora_state
        skey = (tuple(ctx.shape), None if st is None else (st.group_adapter.data_ptr(), st.merged))
        stamp = (ctx.data_ptr(), ctx._version)
        hit = self._kv_cache.get(skey)
        if hit is not None:
            self._kv_cache[skey] = self._kv_cache.pop(skey)      # most recently USED last: eviction takes the idle shape, not the oldest-inserted one
        if hit is not None and hit[0] == stamp:
            return hit[1], hit[2]
        B, Nk, Cx = ctx.shape
        inner = self.inner_dim
        kv_out = hit[4] if hit is not None else None
        vt_out = hit[2] if hit is not None else None
        if self._has_lora():
            k = self.to_k(ctx); v = self.to_v(ctx)
            kv_out = None
        else:
            ms = self._merged()
            x2 = ctx.reshape(B * Nk, Cx)
            if ms is not None:
                kv = ops.gemm(x2, self.kv_slots(), groups=ms.groups, w_group_adapter=ms.group_adapter, out=kv_out)
            else:
                kv = ops.gemm(x2, self.kv_weight(), out=kv_out)
            kv_out = kv
            kv = kv.view(B, Nk, 2 * inner)
            k, v = kv[:, :, :inner], kv[:, :, inner:]
        vt = ops.transpose_v(v, self.heads, out=vt_out)
        if hit is None and len(self._kv_cache) >= 4:
            self._kv_cache.pop(next(iter(self._kv_cache)))
            bump_pointer_epoch()                      # a captured graph may still point at the evicted K / V^T
        self._kv_cache[skey] = (stamp, k, vt, ctx, kv_out)
        return k, vt


def _ip_kv(attn: Attention, ip_ctx: torch.Tensor):
    """K / V^T of the image-prompt tokens for this layer, cached PER SHAPE of ``ip_ctx`` (as ``project_cross`` caches per context shape):
    a hit on (pointer, version) returns the stored projections; a change of content for a shape seen before recomputes INTO the stored
    tensors, so the pointers a captured graph recorded stay valid.  Called by the forward and — before step graphs are replayed on a
    new request — eagerly by ``UNet2DConditionModel.refresh_ip_kv`` (a replay runs no Python: without that refresh the second request
    of a graph-mode InstantID session attended to the FIRST request's identity tokens).  Round 4: the cache held ONE
    entry, so two InstantID engines of different shapes alternating A, B, A re-allocated A's tensors on the third call while A's
    graphs still read the freed ones; now each shape keeps its tensors, and evicting the oldest of more than four bumps the pointer
    epoch (every captured graph is dropped and re-recorded)."""
    shape = tuple(ip_ctx.shape)
    stamp = (ip_ctx.data_ptr(), ip_ctx._version)
    cache = attn._ip_cache
    if cache is None:
        cache = attn._ip_cache = {}
    c = cache.get(shape)
    if c is not None:
        cache[shape] = cache.pop(shape)               # LRU order (see project_cross)
    if c is None or c[0] != stamp:
        Bc, Ni, Cx = shape
        kv_out = c[3] if c is not None else None
        vt_out = c[2] if c is not None else None
        kv = ops.gemm(ip_ctx.reshape(Bc * Ni, Cx), attn.ip_kv_weight, out=kv_out)
        kv3 = kv.view(Bc, Ni, 2 * attn.inner_dim)
        vt = ops.transpose_v(kv3[:, :, attn.inner_dim:], attn.heads, out=vt_out)
        if c is None and len(cache) >= 4:
            cache.pop(next(iter(cache)))
            bump_pointer_epoch()                      # a captured graph may still point at the evicted K / V^T
        c = (stamp, kv3[:, :, :attn.inner_dim], vt, kv, shape, ip_ctx)
        cache[shape] = c
    return c


def _ip_branch(attn: Attention, q: torch.Tensor, o: torch.Tensor, ip_ctx: torch.Tensor, row0: int) -> None:
    """o[row0:] += ip_scale * softmax(q[row0:] K_ip^T) V_ip   (src/ip_adapter/attention_processor.py:391-409)."""
    c = _ip_kv(attn, ip_ctx)
    ops.attention(q[row0:], c[1], c[2], attn.heads, attn.scale, out=o[row0:], accumulate=True, out_scale=attn.ip_scale)


def _to_tokens(attn, hidden_states):
    if isinstance(hidden_states, ops.Mx8Tensor):
        return hidden_states
    if hidden_states.dim() == 4:
        raise L.OmgHipError("4-D (NCHW) hidden_states are not produced by the SDXL transformer blocks; pass (B, N, C)")
    return hidden_states


class FusedAttnProcessor:
    """softmax(QK^T/sqrt(d))V on the flash kernel.  ``residual`` (optional kwarg used by this
    package's own BasicTransformerBlock) is added in the out-projection's epilogue."""

    supports_fused_residual = True

    def _qk_src(self, attn, is_cross: bool, n_tokens: int, batch: int, device, main_batch=None, images=1):
        return None

    def _skip_layer(self) -> None:
        return None

    def __call__(self, attn: Attention, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale: float = 1.0, residual: Optional[torch.Tensor] = None, **cross_attention_kwargs):
        if attention_mask is not None:
            raise L.OmgHipError("attention_mask is not supported")
        x = _to_tokens(attn, hidden_states)
        B, N, _ = x.shape
        is_cross = encoder_hidden_states is not None
        if is_cross:
            q = attn.to_q(x)
            k, vt = attn.project_cross(encoder_hidden_states)
        else:
            q, k, vt = attn.project_self(x)
        bypass = cross_attention_kwargs.pop("omg_bypass_controller", False)
        main_b = cross_attention_kwargs.pop("omg_main_batch", None)       # p2p batch per request [unc0,unc1,cond0,cond1]
        n_img = cross_attention_kwargs.pop("omg_images", 1)               # requests batched in lock-step
        twin = cross_attention_kwargs.pop("omg_twin", False)               # rows are [unc, cond] of requests whose two samples coincide
        if twin and not bypass:
            self._skip_layer()
        src = None if (bypass or twin) else self._qk_src(attn, is_cross, N, B, x.device, main_b, n_img)
        ip_ctx = cross_attention_kwargs.pop("omg_ip_tokens", None)         # (rows, 16, Cx) image-prompt tokens of the rows
        ip_row0 = cross_attention_kwargs.pop("omg_ip_rows", 0)             #   [ip_row0:] of the batch (InstantID concept samples)
        o = ops.attention(q, k, vt, attn.heads, attn.scale, qk_src=src)
        if is_cross and ip_ctx is not None and attn.ip_kv_weight is not None:
            _ip_branch(attn, q, o, ip_ctx, ip_row0)
        out = attn.to_out[0](o, residual=residual)
        return out


class RegionControlNet_AttnProcessor(FusedAttnProcessor):
    """Drop-in for the reference class of the same name (lora_pipeline.py:61-133)."""

    def __init__(self, attention_op=None, controller=None, place_in_unet=None):
        self.attention_op = attention_op
        self.controller = controller
        self.place_in_unet = place_in_unet

    def _fusable(self) -> bool:
        return self.controller is None or getattr(self.controller, "is_pure_replacement", False)

    def _qk_src(self, attn, is_cross, n_tokens, batch, device, main_batch=None, images=1):
        if self.controller is None:
            return None
        return self.controller.fused_qk_src(is_cross, n_tokens, main_batch or batch, self.place_in_unet, device=device,
                                            total_batch=batch, images=images)

    def _skip_layer(self) -> None:
        """The edited sample IS the base sample (pipeline ``dedup``): the replacement is the identity, only the counters move."""
        if self.controller is not None:
            self.controller.skip_layer()

    def __call__(self, attn: Attention, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale: float = 1.0, residual: Optional[torch.Tensor] = None, **cross_attention_kwargs):
        if self._fusable() or cross_attention_kwargs.get("omg_bypass_controller", False):
            return super().__call__(attn, hidden_states, encoder_hidden_states, attention_mask, temb, scale,
                                    residual=residual, **cross_attention_kwargs)
        # ---- protocol mode: the reference's literal sequence (lora_pipeline.py:98-124)
        if cross_attention_kwargs.get("omg_images", 1) != 1:
            raise L.OmgHipError("protocol mode (non-identity mapper) runs one request at a time")
        x = _to_tokens(attn, hidden_states)
        is_cross = encoder_hidden_states is not None
        src = encoder_hidden_states if is_cross else x
        query = attn.head_to_batch_dim(attn.to_q(x))
        key = attn.head_to_batch_dim(attn.to_k(src))
        value = attn.to_v(src)
        probs = attn.get_attention_scores(query, key, None)
        probs = self.controller(probs, is_cross, self.place_in_unet)
        o = ops.attn_apply_probs(probs.contiguous(), value, attn.heads)
        return attn.to_out[0](o, residual=residual)


class IPAttnProcessor2_0(nn.Module):
    """InstantID / IP-Adapter cross-attention (src/ip_adapter/attention_processor.py:296-424):
    the last ``num_tokens`` context rows are image-prompt tokens with their own K/V projections;
    output = text attention + scale * ip attention (second kernel call accumulates)."""

    supports_fused_residual = True

    def __init__(self, hidden_size: int, cross_attention_dim: Optional[int] = None, scale: float = 1.0, num_tokens: int = 4,
                 dtype=torch.float16, device=None):
        super().__init__()
        self.hidden_size, self.cross_attention_dim, self.scale, self.num_tokens = hidden_size, cross_attention_dim, scale, num_tokens
        self.to_k_ip = Linear(cross_attention_dim or hidden_size, hidden_size, bias=False, dtype=dtype, device=device)
        self.to_v_ip = Linear(cross_attention_dim or hidden_size, hidden_size, bias=False, dtype=dtype, device=device)

    def __call__(self, attn: Attention, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 residual: Optional[torch.Tensor] = None, **kw):
        if attention_mask is not None:
            raise L.OmgHipError("attention_mask is not supported")
        x = _to_tokens(attn, hidden_states)
        if encoder_hidden_states is None:
            q, k, vt = attn.project_self(x)
            o = ops.attention(q, k, vt, attn.heads, attn.scale)
            return attn.to_out[0](o, residual=residual)
        end = encoder_hidden_states.shape[1] - self.num_tokens
        text, ip = encoder_hidden_states[:, :end], encoder_hidden_states[:, end:]
        q = attn.to_q(x)
        B, Nt, Cx = text.shape
        k = attn.to_k(text.reshape(B * Nt, Cx)).view(B, Nt, -1)
        v = attn.to_v(text.reshape(B * Nt, Cx)).view(B, Nt, -1)
        o = ops.attention(q, k, ops.transpose_v(v, attn.heads), attn.heads, attn.scale)
        Ni = ip.shape[1]
        ipc = ip.reshape(B * Ni, Cx)
        k_ip = self.to_k_ip(ipc).view(B, Ni, -1)
        v_ip = self.to_v_ip(ipc).view(B, Ni, -1)
        ops.attention(q, k_ip, ops.transpose_v(v_ip, attn.heads), attn.heads, attn.scale, out=o, accumulate=True,
                      out_scale=self.scale)
        return attn.to_out[0](o, residual=residual)
