"""ORACLE (test infrastructure): CPU restatement of the block-scaled fp8 ("MX", OCP Microscaling Formats v1.0) operand format
of omg_quant_mx8 / omg_gemm_mx8.

The reference (kongzhecn/OMG) has no fp8 path — it runs fp16 everywhere (inference_lora.py:153-159); the fp8 GEMM exists because
BASELINE.json's north_star / configs[4] ask for it.  There is therefore no reference behaviour to pin: PARITY UNPINNED by the
reference.  What this file is pinned against instead: the element cast is ``torch.Tensor.to(torch.float8_e4m3fn)`` (PyTorch's own
OCP e4m3 round-to-nearest-even conversion), and the dequantised product is a plain fp32 matmul.

Format: elements e4m3 (bias 7, max 448, no infinities), one shared E8M0 scale 2^(s - 127) per 32 consecutive K elements.
Scale rule of this package: the smallest power of two with amax / scale <= 448, i.e. no element saturates (the OCP document's
example rule floor(log2 amax) - 8 would clamp amax in (448, 512) x 2^e).  Scale layout: uint32 S[K/128][rows], byte b of
S[t][r] = scale of row r, K block 4t + b.
"""
from __future__ import annotations

import torch

E4M3_MAX = 448.0


def block_exponents(x: torch.Tensor) -> torch.Tensor:
    """x [rows, K] fp32 -> int32 [rows, K/32]: e with scale = 2^e (biased byte = e + 127)."""
    rows, K = x.shape
    amax = x.abs().reshape(rows, K // 32, 32).amax(dim=-1)
    v = amax * torch.tensor(1.0 / 448.0, dtype=torch.float32)          # the kernel multiplies by the fp32 reciprocal
    m, e = torch.frexp(v)                                              # v = m * 2^e, m in [0.5, 1)
    ex = torch.where(m == 0.5, e - 1, e).to(torch.int32)               # ceil(log2 v)
    ex = torch.where(v == 0, torch.full_like(ex, -127), ex)
    return ex.clamp(-127, 127)


def quantize(x: torch.Tensor):
    """-> (q uint8 [rows, K] (e4m3 bit patterns), scales int32 [K/128, rows] packed as the kernel packs them, exps [rows, K/32])."""
    x = x.float()
    rows, K = x.shape
    assert K % 128 == 0
    ex = block_exponents(x)
    inv = torch.ldexp(torch.ones((), dtype=torch.float32), -ex)        # exact powers of two
    y = (x.reshape(rows, K // 32, 32) * inv[..., None]).reshape(rows, K)
    q = y.to(torch.float8_e4m3fn).view(torch.uint8)
    b = (ex + 127).to(torch.int64).reshape(rows, K // 128, 4)
    packed = (b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16) | (b[..., 3] << 24))
    packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)
    return q, packed.t().contiguous(), ex


def dequantize(q: torch.Tensor, ex: torch.Tensor) -> torch.Tensor:
    rows, K = q.shape
    v = q.view(torch.float8_e4m3fn).float().reshape(rows, K // 32, 32)
    return (v * torch.ldexp(torch.ones((), dtype=torch.float32), ex)[..., None]).reshape(rows, K)


def unpack_scales(packed: torch.Tensor, rows: int) -> torch.Tensor:
    """int32 [K/128, s_ld] -> exponents int32 [rows, K/32]."""
    p = packed[:, :rows].t().to(torch.int64) & 0xFFFFFFFF
    b = torch.stack([(p >> (8 * i)) & 0xFF for i in range(4)], dim=-1)           # [rows, K/128, 4]
    return (b.reshape(rows, -1) - 127).to(torch.int32)
