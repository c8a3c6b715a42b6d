"""Oracle parity AT THE SIZES THE BENCHMARK RUNS (-m gpu) — VERDICT r1 "next" item 1.

Every (kernel variant, shape) row of profiles/r01_by_shape_v9.txt above ~1 % of a bench step is launched here with its real
extents (M up to 1,048,576 rows, 64 per-sample weight slots, conv K up to 23,040, attention (64, 10, 4096, 4096)) through the
C ABI, and a few thousand SAMPLED output rows are compared with an fp32 dot product computed ON THE CPU from the same
16-bit-rounded operands (torch CPU matmul: independent of every kernel in this package).  A dropped K-tile, a dropped conv
tap, a wrong weight slot or a wrapped 32-bit buffer offset changes the sampled rows by O(1); the tolerance is 16-bit
rounding of an O(1) result.  Row samples always include the first and last rows of the matrix, both sides of tile and
sample boundaries, and the rows nearest to the 2 GiB buffer-offset guard.

Also here: one full-width (C = 1280, 20 heads, 32 x 32 tokens) BasicTransformerBlock and one full-width ResnetBlock2D
against oracle/unet.py.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from omg_amd import _lib as L
from omg_amd import ops
from oracle import unet as ou

DT = torch.float16
RTOL, ATOL = 2e-3, 3e-3          # fp16 storage of O(1) results accumulated in fp32


def gen(shape, dev, seed, scale=1.0, dtype=DT):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * scale).to(dtype)


def sample_rows(M, n, dev, seed, marks=()):
    """n random rows + the matrix corners + both sides of every boundary in `marks`."""
    g = torch.Generator().manual_seed(seed)
    rows = set(torch.randint(0, M, (n,), generator=g).tolist())
    for m in (0, 1, 255, 256, M - 257, M - 256, M - 2, M - 1) + tuple(marks):
        for d in (-1, 0, 1):
            if 0 <= m + d < M:
                rows.add(m + d)
    return torch.tensor(sorted(rows), dtype=torch.long)


def check(out_rows, ref, what):
    out_rows = out_rows.float().cpu()
    err = (out_rows - ref).abs()
    tol = ATOL + RTOL * ref.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} sampled outputs off, max |d| {err.max().item():.3e} (ref rms {ref.pow(2).mean().sqrt().item():.3f})"
    return err.max().item()


# ------------------------------------------------------------------ Linear layers (per-sample weight slots, epilogues)
LIN = [  # (M, N, K, groups, act, residual)  -- tags of r01_by_shape_v9.txt
    (65536, 10240, 1280, 64, "geglu", False),      # FF-GEGLU at 32x32, 16.7 % of the step
    (65536, 1280, 1280, 64, "none", True),         # out-projections with residual, 7.8 %
    (65536, 1280, 5120, 64, "none", True),         # FF-out, 7.6 %
    (65536, 3840, 1280, 64, "none", False),        # fused q|k|v, 6.2 %
    (32768, 10240, 1280, 1, "geglu", False),       # plain steps (no slots), 3.9 %
    (262144, 5120, 640, 64, "geglu", False),       # FF-GEGLU at 64x64, 3.4 %; C is 1.34 GB
    (262144, 640, 640, 64, "none", True),          # 64x64 projections, the slowest big shape
    (262144, 1920, 640, 64, "none", False),        # q|k|v at 64x64
    (32768, 1280, 1280, 1, "none", True),
]


@pytest.mark.parametrize("M,N,K,groups,act,res", LIN)
def test_linear_at_bench_size_matches_cpu_fp32_on_sampled_rows(dev, M, N, K, groups, act, res):
    a = gen((M, K), dev, 1)
    n_slots = 3 if groups > 1 else 1                                   # [base, concept 1, concept 2] merged stacks
    w = gen((n_slots, N, K), dev, 2, scale=K ** -0.5)
    bias = gen((N,), dev, 3)
    resid = gen((M, N), dev, 4) if res else None
    slot_of_group = None
    kw = {}
    wk = w
    bk = bias
    if act == "geglu":
        perm = ops.geglu_row_perm(N).to(dev)
        wk, bk = w[:, perm].contiguous(), bias[perm].contiguous()
    if groups > 1:
        # the benchmark's layout: per request 4 main samples on slot 0 then [c1, c1, c2, c2] — here shuffled per group
        slot_of_group = torch.tensor([(0, 0, 0, 0, 1, 1, 2, 2)[g % 8] for g in range(groups)], dtype=torch.int32, device=dev)
        kw = dict(groups=groups, w_group_adapter=slot_of_group)
    else:
        wk = wk[0]
    out = ops.gemm(a, wk, bias=bk, residual=resid, act=L.ACT_GEGLU if act == "geglu" else L.ACT_NONE, **kw)
    torch.cuda.synchronize()
    rpg = M // groups
    budget = int(6e10 / (2.0 * N * K))                                 # ~60 GFLOP of CPU fp32 per case
    rows = sample_rows(M, max(256, min(4096, budget)), dev, 5, marks=(rpg, M - rpg, M // 2, (M // 2 // rpg) * rpg))
    a_s = a[rows.to(dev)].float().cpu()
    w_cpu = w.float().cpu()
    b_cpu = bias.float().cpu()
    ref = torch.empty(len(rows), N)
    slots = (rows // rpg).apply_(lambda g: (0, 0, 0, 0, 1, 1, 2, 2)[g % 8]) if groups > 1 else torch.zeros(len(rows), dtype=torch.long)
    for s in range(n_slots):
        sel = slots == s
        if sel.any():
            ref[sel] = a_s[sel] @ w_cpu[s].T + b_cpu
    if act == "geglu":
        val, gate = ref.chunk(2, dim=-1)
        ref = val * F.gelu(gate)
    if res:
        ref = ref + resid[rows.to(dev)].float().cpu()
    worst = check(out[rows.to(dev)], ref, f"lin {M}x{N}x{K} g{groups} {act}")
    print(f"lin {M}x{N}x{K} groups {groups} {act}{' +res' if res else ''}: {len(rows)} rows, max |d| {worst:.2e}")


# ------------------------------------------------------------------ implicit-GEMM convolutions
CONV = [  # (B, H, C1, C2, Cout, stride, upsample)
    (64, 32, 1280, 0, 1280, 1, False),       # conv (65536, 1280, 11520), 3.0 %
    (64, 128, 320, 0, 320, 1, False),        # conv (1048576, 320, 2880): M = 1,048,576 rows, 128x320 tile (v24)
    (64, 64, 640, 0, 640, 1, False),         # conv (262144, 640, 5760)
    (64, 32, 1280, 1280, 1280, 1, False),    # conv (65536, 1280, 23040): concat input, K = 23,040
    (64, 128, 640, 0, 320, 1, False),        # conv (1048576, 320, 5760)
    (64, 64, 640, 0, 640, 1, True),          # upsample folded into the loader: conv (1048576, 640, 5760), A and C 1.34 GB
    (64, 128, 320, 0, 320, 2, False),        # stride-2 downsampler
]


@pytest.mark.parametrize("B,H,C1,C2,Co,stride,ups", CONV)
def test_conv_at_bench_size_matches_cpu_fp32_on_sampled_pixels(dev, B, H, C1, C2, Co, stride, ups):
    x1 = gen((B, H, H, C1), dev, 11)
    x2 = gen((B, H, H, C2), dev, 12) if C2 else None
    Ct = C1 + C2
    w_oihw = gen((Co, Ct, 3, 3), dev, 13, scale=(9 * Ct) ** -0.5)
    bias = gen((Co,), dev, 14)
    gb = gen((B, Co), dev, 15)                                         # the time-embedding projection (per-sample bias)
    y = ops.conv2d(x1, ops.pack_conv_weight(w_oihw), 3, stride=stride, upsample=ups, x2=x2, bias=bias, group_bias=gb)
    torch.cuda.synchronize()
    Hl = 2 * H if ups else H
    Ho = (Hl + 2 - 3) // stride + 1
    assert y.shape == (B, Ho, Ho, Co)
    M = B * Ho * Ho
    budget = int(4e10 / (2.0 * Co * 9 * Ct))
    rows = sample_rows(M, max(256, min(4096, budget)), dev, 16, marks=(Ho * Ho, M - Ho * Ho, Ho, Ho * Ho - Ho, M // 2))
    b_i, rem = rows // (Ho * Ho), rows % (Ho * Ho)
    oy, ox = rem // Ho, rem % Ho
    # gather the 3x3 patches of the sampled output pixels on the device (data movement only), dot products on the CPU
    xin = x1 if x2 is None else None
    patches = torch.zeros(len(rows), 3, 3, Ct, dtype=torch.float32)
    bi_d = b_i.to(dev)
    for dy in range(3):
        for dx in range(3):
            iy, ix = oy * stride + dy - 1, ox * stride + dx - 1
            ok = (iy >= 0) & (iy < Hl) & (ix >= 0) & (ix < Hl)
            sy, sx = (iy // 2, ix // 2) if ups else (iy, ix)
            sy, sx = sy.clamp(0, H - 1).to(dev), sx.clamp(0, H - 1).to(dev)
            v = x1[bi_d, sy, sx].float().cpu()
            if x2 is not None:
                v = torch.cat([v, x2[bi_d, sy, sx].float().cpu()], dim=-1)
            patches[:, dy, dx] = v * ok[:, None].float()
    w_cpu = w_oihw.float().cpu().permute(0, 2, 3, 1).reshape(Co, 9 * Ct)
    ref = patches.reshape(len(rows), 9 * Ct) @ w_cpu.T + bias.float().cpu() + gb.float().cpu()[b_i]
    worst = check(y.view(M, Co)[rows.to(dev)], ref, f"conv B{B} {H}x{H} {C1}+{C2}->{Co} s{stride} up{int(ups)}")
    print(f"conv M={M} N={Co} K={9 * Ct} stride {stride} upsample {int(ups)}: {len(rows)} pixels, max |d| {worst:.2e}")


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,heads,Nq,Nkv,borrow", [(64, 10, 4096, 4096, True), (64, 20, 1024, 1024, True), (64, 20, 1024, 77, True),
                                                   (64, 10, 4096, 77, False), (32, 10, 4096, 4096, False)])
def test_attention_at_bench_size_matches_cpu_fp32_on_sampled_queries(dev, B, heads, Nq, Nkv, borrow):
    C = heads * 64
    q = gen((B, Nq, C), dev, 21)
    k = gen((B, Nkv, C), dev, 22, scale=1.5)          # logits with a realistic spread (row entropy well below uniform)
    v = gen((B, Nkv, C), dev, 23)
    src = None
    if borrow:                                         # 8 requests x [unc0, unc1, cond0, cond1 | 4 concept samples]: cond1 borrows cond0's Q, K
        idx = list(range(B))
        for r in range(B // 8):
            idx[8 * r + 3] = 8 * r + 2
        src = torch.tensor(idx, dtype=torch.int32, device=dev)
    o = ops.attention(q, k, ops.transpose_v(v, heads), heads, 0.125, qk_src=src)
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(24)
    picks = [(0, 0), (B - 1, heads - 1), (3, 1), (2, 1)] + [(int(torch.randint(0, B, (1,), generator=g)), int(torch.randint(0, heads, (1,), generator=g))) for _ in range(8)]
    worst = 0.0
    for (b, h) in picks:
        rows = sample_rows(Nq, 96, dev, 25 + b, marks=(64, 128, Nq - 64))
        bs = int(src[b]) if src is not None else b
        qs = q[bs, rows.to(dev), h * 64:(h + 1) * 64].float().cpu()
        ks = k[bs, :, h * 64:(h + 1) * 64].float().cpu()
        vs = v[b, :, h * 64:(h + 1) * 64].float().cpu()
        ref = torch.softmax(qs @ ks.T * 0.125, dim=-1) @ vs
        worst = max(worst, check(o[b, rows.to(dev), h * 64:(h + 1) * 64], ref, f"attn ({B},{heads},{Nq},{Nkv}) sample {b} head {h}"))
    print(f"attn ({B},{heads},{Nq},{Nkv}) borrow={borrow}: {len(picks)} (sample, head) pairs x ~100 query rows, max |d| {worst:.2e}")


# ------------------------------------------------------------------ full-width modules vs oracle/unet.py
def _oracle_sd(module, prefix):
    return {f"{prefix}.{k}": v.detach().float().cpu() for k, v in module.state_dict().items()}


def _init(module, dev, seed, qk_gain=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in module.named_parameters():
        if name.endswith(".weight") and p.dim() >= 2:
            w = torch.randn(p.shape, generator=g, device=dev) * p[0].numel() ** -0.5
        elif name.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g, device=dev)
        else:
            w = 0.1 * torch.randn(p.shape, generator=g, device=dev)
        p.data.copy_(w.to(p.dtype))


@pytest.mark.parametrize("dtype,tol,N", [(torch.float16, 1.5e-2, 1024), (torch.bfloat16, 8e-2, 1024), (torch.float16, 1.5e-2, 1008)])
def test_full_width_transformer_block_matches_oracle(dev, dtype, tol, N):
    """C = 1280, 20 heads, 32 x 32 tokens, context (77, 2048): the block that runs 60 times per SDXL forward.  N = 1008 = 36 x 28: the deepest
    level of a 1152 x 896 image (B3's height / width, lora_pipeline.py:217-218) — a ragged last key tile and a ragged query block in the
    self-attention, a row count that is no multiple of 256 in every Linear."""
    from omg_amd.unet import BasicTransformerBlock
    blk = BasicTransformerBlock(1280, 20, 2048, dtype, dev)
    _init(blk, dev, 31)
    B = 2
    x = gen((B, N, 1280), dev, 32, dtype=dtype)
    ctx = gen((B, 77, 2048), dev, 33, dtype=dtype)
    y = blk(x, ctx, {}).float().cpu()
    sd = _oracle_sd(blk, "b")
    ref = ou.transformer_block(sd, "b", 20, x.float().cpu(), ctx.float().cpu(), ou.plain_attention)
    rms = ref.pow(2).mean().sqrt().item()
    err = (y - ref).abs().max().item() / rms
    print(f"full-width BasicTransformerBlock {dtype}: max |d| / rms = {err:.2e} (rms {rms:.3f})")
    assert err < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1.5e-2), (torch.bfloat16, 8e-2)])
@pytest.mark.parametrize("cin,cskip,cout,H", [(1280, 0, 1280, 32), (1280, 640, 640, 64)])
def test_full_width_resnet_block_matches_oracle(dev, dtype, tol, cin, cskip, cout, H):
    """ResnetBlock2D at 1280 channels / 32 x 32, and the concat (1280 + 640 -> 640) up-block form with its 1x1 shortcut."""
    from omg_amd import ops as O
    from omg_amd.unet import ResnetBlock2D, _Ctx
    ocfg = ou.UNetConfig.sdxl()
    res = ResnetBlock2D(cin + cskip, cout, 1280, 32, 1e-5, dtype, dev)
    _init(res, dev, 41)
    B = 2
    x = gen((B, H, H, cin), dev, 42, dtype=dtype)
    x2 = gen((B, H, H, cskip), dev, 43, dtype=dtype) if cskip else None
    temb = gen((B, 1280), dev, 44, dtype=dtype)
    y = res(x, _Ctx(O.silu(temb), None, B), x2=x2).float().cpu()
    sd = _oracle_sd(res, "r")
    xin = x if x2 is None else torch.cat([x, x2], dim=-1)
    ref = ou.resnet_block(sd, "r", ocfg, xin.float().cpu().permute(0, 3, 1, 2), temb.float().cpu()).permute(0, 2, 3, 1)
    rms = ref.pow(2).mean().sqrt().item()
    err = (y - ref).abs().max().item() / rms
    print(f"full-width ResnetBlock2D {cin}+{cskip}->{cout} @{H} {dtype}: max |d| / rms = {err:.2e} (rms {rms:.3f})")
    assert err < tol
