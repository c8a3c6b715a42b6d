"""MX-fp8 operands (-m gpu): omg_quant_mx8 and omg_gemm_mx8 against oracle/mx8.py.

* quantiser: the e4m3 bytes and the packed E8M0 scale dwords are compared BIT FOR BIT with the oracle (element cast =
  torch's own float8_e4m3fn conversion);
* GEMM: compared with an fp32 CPU matmul of the DEQUANTISED operands — the kernel's arithmetic is then exact products
  accumulated in fp32, so the tolerance is fp32 accumulation order + the 16-bit store, not "fp8 accuracy";
* precision report: the same layer in fp16 and in MX-fp8 against the fp32 product of the un-quantised operands.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from omg_amd import _lib as L
from omg_amd import ops
from oracle import mx8


def gen(shape, seed, scale=1.0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K", [(64, 128), (300, 1280), (37, 256), (4096, 640)])
def test_quantiser_is_bit_exact(dev, dtype, M, K):
    x = gen((M, K), 1, dtype=dtype)
    x[0, :32] = 0                                   # an all-zero block
    x[1 % M, 33] = 448.0                            # amax exactly on the e4m3 maximum
    x[2 % M, 64:96] *= 1e-3                         # a small-magnitude block next to normal ones
    x[3 % M, 100] = 6.0e4 if dtype == torch.float16 else 3.0e38      # near the top of the input format
    got = ops.quant_mx8(x.to(dev))
    q, packed, ex = mx8.quantize(x.float())
    assert torch.equal(got.q.cpu(), q), f"{(got.q.cpu() != q).sum().item()} element bytes differ"
    assert torch.equal(got.scales.cpu()[:, :M], packed)


def _mx(x, dev):
    t = ops.quant_mx8(x.to(dev))
    q, packed, ex = mx8.quantize(x.float())
    return t, mx8.dequantize(q, ex)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (256, 256, 256), (300, 136, 384), (1024, 1280, 1280), (77 * 4, 640, 2048), (520, 264, 5120)])
def test_gemm_mx8_matches_dequantised_fp32(dev, dtype, M, N, K):
    a, w = gen((M, K), 2, dtype=dtype), gen((N, K), 3, scale=K ** -0.5, dtype=dtype)
    a[:, 5] *= 30.0                                  # an outlier channel: block scales differ along K
    bias, res = gen((N,), 4, dtype=dtype), gen((M, N), 5, dtype=dtype)
    ta, da = _mx(a, dev)
    tw, dw = _mx(w, dev)
    out = ops.gemm_mx8(ta, tw, out_dtype=dtype, bias=bias.to(dev), residual=res.to(dev), out_scale=0.5)
    ref = (da @ dw.T + bias.float()) * 0.5 + res.float()
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1.6e-2, atol=1.6e-2)
    torch.testing.assert_close(out.float().cpu(), ref, **tol)
    out2 = ops.gemm_mx8(ta, tw, out_dtype=dtype, act=L.ACT_SILU)
    torch.testing.assert_close(out2.float().cpu(), F.silu(da @ dw.T), **tol)


def test_gemm_mx8_asymmetric_identity(dev):
    """A = I with an asymmetric W catches a transposed tile or a mis-ordered K byte (every W element is exactly representable)."""
    n = 256
    a = torch.eye(n, dtype=torch.float16)
    w = ((torch.arange(n * n).reshape(n, n) % 15) - 7).to(torch.float16)      # small integers: exact in e4m3 at scale 2^-5..
    ta, _ = _mx(a, dev)
    tw, dw = _mx(w, dev)
    assert torch.equal(dw, w.float())
    out = ops.gemm_mx8(ta, tw)
    assert torch.equal(out.float().cpu(), w.float().T)


def test_gemm_mx8_geglu_and_weight_slots(dev):
    """GEGLU epilogue on row-interleaved weights, and per-sample weight slots (merged-LoRA mode) with their scale columns."""
    dtype = torch.float16
    B, rows, C = 6, 512, 640
    N = 8 * C
    a = gen((B * rows, C), 6, dtype=dtype)
    w = gen((3, N, C), 7, scale=C ** -0.5, dtype=dtype)
    b = gen((N,), 8, dtype=dtype)
    perm = ops.geglu_row_perm(N)
    ta, da = _mx(a, dev)
    tw, dwp = _mx(w[:, perm].reshape(3 * N, C), dev)
    slot = torch.tensor([0, 2, 1, 1, 0, 2], dtype=torch.int32)
    out = ops.gemm_mx8(ta, tw, bias=b[perm].contiguous().to(dev), act=L.ACT_GEGLU, groups=B, w_group_adapter=slot.to(dev), n_per_adapter=N)
    inv = torch.argsort(perm)
    dw = dwp.reshape(3, N, C)[:, inv]                # back to natural row order
    ref = torch.empty(B * rows, N // 2)
    for g in range(B):
        h = da[g * rows:(g + 1) * rows] @ dw[slot[g]].T + b.float()
        val, gate = h.chunk(2, dim=-1)
        ref[g * rows:(g + 1) * rows] = val * F.gelu(gate)
    torch.testing.assert_close(out.float().cpu(), ref, rtol=3e-3, atol=4e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,kind", [(2048, 1280, 1280, "res"), (1500, 520, 128, "bias"), (4096, 2560, 640, "geglu"), (3 * 1024, 1280, 256, "slots"),
                                        (8192, 1024, 384, "silu")])
def test_persistent_mx8_gemm_is_bitwise_the_one_tile_kernel(dev, dtype, M, N, K, kind):
    """gemm_mx8_kernel_p (round 6: a persistent tile walk with the next tile's stage 0 issued under the current tile's last k-step; the two stage buffers
    swap roles from tile to tile when the stage count is even) runs the loads, the MFMA order and the epilogue of the one-tile-per-block kernel: every
    epilogue form, one / three / ten stages per tile (K = 128 / 384 / 1280: odd and even counts, i.e. with and without the buffer swap), ragged M and N,
    per-sample weight slots with a skipped group — with one block per tile, with the CU count, and with EIGHT blocks so that every block walks many tiles."""
    lib = L.lib()
    a = gen((M, K), 21, dtype=dtype)
    ta = ops.quant_mx8(a.to(dev))
    kw = {}
    if kind == "slots":
        w = gen((3, N, K), 22, scale=K ** -0.5, dtype=dtype)
        tw = ops.quant_mx8(w.reshape(3 * N, K).to(dev))
        kw = dict(groups=3, w_group_adapter=torch.tensor([2, -1, 0], dtype=torch.int32, device=dev), n_per_adapter=N, bias=gen((N,), 23, dtype=dtype).to(dev))
    else:
        w = gen((N, K), 22, scale=K ** -0.5, dtype=dtype)
        bias = gen((N,), 23, dtype=dtype).to(dev)
        if kind == "geglu":
            perm = ops.geglu_row_perm(N)
            tw = ops.quant_mx8(w[perm].contiguous().to(dev))
            kw = dict(bias=bias[perm.to(dev)].contiguous(), act=L.ACT_GEGLU)
        else:
            tw = ops.quant_mx8(w.to(dev))
            kw = dict(bias=bias)
            if kind == "res":
                kw.update(residual=gen((M, N), 24, dtype=dtype).to(dev), out_scale=0.5)
            if kind == "silu":
                kw.update(act=L.ACT_SILU)
    outs = {}
    try:
        for name, word in (("one_tile", 12 | (128 << 8)), ("persistent", 12 | (512 << 8)), ("eight_blocks", 12 | ((256 | 512) << 8))):      # 512: the walk whatever the shape
            lib.omg_debug_set_mx8_split(word)
            o = torch.full((M, N // 2 if kind == "geglu" else N), float("nan"), dtype=dtype, device=dev)
            ops.gemm_mx8(ta, tw, out=o, out_dtype=dtype, **kw)
            outs[name] = o
    finally:
        lib.omg_debug_set_mx8_split(12)
    if kind == "slots":      # the skipped group's rows are never written by either form
        assert torch.isnan(outs["one_tile"][1024:2048]).all() and torch.isnan(outs["persistent"][1024:2048]).all()
        outs = {k_: torch.cat([v[:1024], v[2048:]]) for k_, v in outs.items()}
    assert torch.isfinite(outs["one_tile"]).all()
    assert torch.equal(outs["persistent"], outs["one_tile"]) and torch.equal(outs["eight_blocks"], outs["one_tile"])


def test_gemm_mx8_at_bench_size_on_sampled_rows(dev):
    """(65536, 10240, 1280) with 64 weight-slot groups and GEGLU, (65536, 1280, 5120) with residual: sampled rows vs CPU fp32."""
    for (M, N, K, act) in [(65536, 10240, 1280, "geglu"), (65536, 1280, 5120, "none")]:
        g = torch.Generator(device=dev).manual_seed(9)
        a = torch.randn((M, K), generator=g, device=dev).half()
        w = (torch.randn((3, N, K), generator=g, device=dev) * K ** -0.5).half()
        bias = torch.randn((N,), generator=g, device=dev).half()
        resid = torch.randn((M, N), generator=g, device=dev).half() if act == "none" else None
        wk, bk = w, bias
        if act == "geglu":
            perm = ops.geglu_row_perm(N).to(dev)
            wk, bk = w[:, perm].contiguous(), bias[perm].contiguous()
        ta, tw = ops.quant_mx8(a), ops.quant_mx8(wk.reshape(3 * N, K))
        slot = torch.tensor([(0, 0, 0, 0, 1, 1, 2, 2)[i % 8] for i in range(64)], dtype=torch.int32, device=dev)
        out = ops.gemm_mx8(ta, tw, bias=bk, residual=resid, act=L.ACT_GEGLU if act == "geglu" else L.ACT_NONE, groups=64,
                           w_group_adapter=slot, n_per_adapter=N)
        rows = torch.cat([torch.tensor([0, 1, 255, 256, 1023, 1024, M - 257, M - 256, M - 1]), torch.randint(0, M, (1500,), generator=torch.Generator().manual_seed(1))]).unique()
        da = mx8.dequantize(ta.q[rows.to(dev)].cpu(), mx8.unpack_scales(ta.scales.cpu(), M)[rows])
        exw = mx8.unpack_scales(tw.scales.cpu(), 3 * N)
        dw = mx8.dequantize(tw.q.cpu(), exw).reshape(3, N, K)
        if act == "geglu":
            dw = dw[:, torch.argsort(perm.cpu())]
        ref = torch.empty(len(rows), N)
        sl = (rows // (M // 64)).apply_(lambda i: (0, 0, 0, 0, 1, 1, 2, 2)[i % 8])
        for s in range(3):
            sel = sl == s
            ref[sel] = da[sel] @ dw[s].T + bias.float().cpu()
        if act == "geglu":
            val, gate = ref.chunk(2, dim=-1)
            ref = val * F.gelu(gate)
        else:
            ref = ref + resid[rows.to(dev)].float().cpu()
        err = (out[rows.to(dev)].float().cpu() - ref).abs()
        assert (err <= 3e-3 + 2e-3 * ref.abs()).all(), (M, N, K, err.max().item())
        print(f"mx8 {M}x{N}x{K} {act}: {len(rows)} rows, max |d| vs dequantised fp32 {err.max().item():.2e}")


def test_precision_report_fp16_vs_mx8(dev):
    """Per-precision error of one FF-GEGLU-sized layer against the fp32 product of the UN-quantised operands (printed; the bound is
    the e4m3 element rounding 2^-4 averaged over K, far above fp16's)."""
    M, N, K = 2048, 2560, 1280
    a, w = gen((M, K), 11), gen((N, K), 12, scale=K ** -0.5)
    a[:, 7] *= 20.0
    ref = a.float() @ w.float().T
    y16 = ops.gemm(a.to(dev), w.to(dev)).float().cpu()
    y8 = ops.gemm_mx8(ops.quant_mx8(a.to(dev)), ops.quant_mx8(w.to(dev))).float().cpu()
    rms = ref.pow(2).mean().sqrt().item()
    e16, e8 = (y16 - ref).pow(2).mean().sqrt().item() / rms, (y8 - ref).pow(2).mean().sqrt().item() / rms
    print(f"relative rms error vs fp32: fp16 path {e16:.2e}, MX-fp8 path {e8:.2e} (max |d| / rms {(y8 - ref).abs().max().item() / rms:.2e})")
    assert e16 < 1e-3 and e8 < 6e-2


def test_layernorm_mx8_equals_layernorm_then_quantise(dev):
    """The fused LayerNorm -> MX-fp8 kernel rounds to the 16-bit value omg_layernorm would have stored before quantising, so it
    must equal the two-kernel sequence bit for bit."""
    for dtype in (torch.float16, torch.bfloat16):
        for (M, C) in [(300, 1280), (1030, 640), (77, 128)]:
            x = gen((M, C), 21, scale=3.0, dtype=dtype).to(dev)
            g, b = gen((C,), 22, dtype=dtype).to(dev) + 1, gen((C,), 23, dtype=dtype).to(dev)
            fused = ops.layernorm_mx8(x, g, b, 1e-5)
            two = ops.quant_mx8(ops.layernorm(x, g, b, 1e-5))
            assert torch.equal(fused.q, two.q) and torch.equal(fused.scales[:, :M], two.scales[:, :M])


@pytest.mark.parametrize("lora", [False, True])
def test_unet_in_mx8_mode_vs_fp16_mode_and_oracle(dev, lora):
    """Module-level Δ of the MX-fp8 Linear path (SURVEY §7.1 P5 'tolerance report per precision'): the tiny SDXL-topology UNet with
    widths raised to multiples of 128 so that every transformer Linear qualifies; fp16 path and MX-fp8 path vs the fp32 oracle."""
    from omg_amd.unet import UNet2DConditionModel, UNetConfig
    from omg_amd.lora import LoraAdapter, LoraBank
    from omg_amd.pipeline import ConceptModels
    from oracle import unet as ou
    kw = dict(sample_size=16, block_out_channels=(128, 256, 512), transformer_layers_per_block=(1, 1, 2), attention_head_dim=(2, 4, 8),
              cross_attention_dim=128, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32)
    cfg, ocfg = UNetConfig(**kw), ou.UNetConfig(**kw)
    dtype = torch.float16
    sd = ou.init_state_dict(ocfg, seed=0, dtype=dtype)
    unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev)
    unet.load_state_dict({k: v.to(dtype) for k, v in sd.items()})
    B = 4
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 4, 16, 16, generator=g)
    ctx = torch.randn(B, 77, 128, generator=g).to(dtype).float()
    te = torch.randn(B, 64, generator=g).to(dtype).float()
    tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]]).repeat(B, 1)
    olora, state = None, None
    if lora:
        names = ou.lora_target_names(ocfg)
        ow, olora = ou.make_lora(ocfg, names, rank=8, seed=100, scale=0.8, dtype=dtype)
        bank = LoraBank(unet, [LoraAdapter("c0", {k: (a.to(dev), b.to(dev)) for k, (a, b) in ow.items()})])
        bank.build([(("c0", 1.0),)], scale=0.8, mode="merged")
        state = ConceptModels(unet, bank).lora_state([1] * B, merged=True)
    ref = ou.unet_forward(sd, ocfg, x, 500.0, ctx, te, tid, lora=olora)
    out = {}
    n_conv = 0
    for mode, lin, conv in (("fp16", "fp16", "fp16"), ("mx8", "mx8", "fp16"), ("mx8+conv", "mx8", "mx8")):
        unet.set_linear_precision(lin)
        unet.set_conv_precision(conv)
        unet.set_lora_state(state)
        out[mode] = unet(x.to(dev), 500.0, encoder_hidden_states=ctx.to(dev).to(dtype),
                         added_cond_kwargs={"text_embeds": te.to(dev).to(dtype), "time_ids": tid.to(dev)})[0].float().cpu()
        unet.set_lora_state(None)
        if conv == "mx8":
            n_conv = sum(1 for m in unet.modules() if hasattr(m, "mx8_ok") and m.mx8_ok())
    unet.set_linear_precision("fp16")
    unet.set_conv_precision("fp16")
    rms = ref.pow(2).mean().sqrt().item()
    e16 = (out["fp16"] - ref).abs().max().item() / rms
    e8 = (out["mx8"] - ref).abs().max().item() / rms
    e8r = (out["mx8"] - ref).pow(2).mean().sqrt().item() / rms
    e8c = (out["mx8+conv"] - ref).abs().max().item() / rms
    e8cr = (out["mx8+conv"] - ref).pow(2).mean().sqrt().item() / rms
    print(f"UNet forward (lora={lora}) vs fp32 oracle, max |d| / rms: fp16 path {e16:.2e}, MX-fp8 Linear path {e8:.2e} (rms error {e8r:.2e}), "
          f"MX-fp8 Linear + {n_conv} resnet convolutions {e8c:.2e} (rms error {e8cr:.2e}); "
          f"fp8 vs fp16 path {(out['mx8'] - out['fp16']).abs().max().item() / rms:.2e}")
    assert e16 < 3e-2
    assert e8 < 0.6 and e8r < 0.13       # measured 0.38-0.49 and 0.089-0.101: profiles/r02_mx8_precision.txt
    assert e8c < 0.7 and e8cr < 0.16
    assert n_conv >= 10
    assert not torch.equal(out["mx8"], out["fp16"]) and not torch.equal(out["mx8+conv"], out["mx8"])


# ---- MX-fp8 convolution path: GroupNorm -> MX-fp8 feature map -> omg_conv2d_mx8 ------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C1,C2,groups", [(2, 8, 8, 128, 0, 32), (3, 5, 7, 128, 128, 32), (2, 16, 16, 640, 0, 32), (1, 6, 6, 1280, 1280, 32),
                                                   (2, 9, 9, 320, 0, 32), (2, 6, 10, 640, 320, 32)])
def test_groupnorm_mx8_is_the_quantised_groupnorm(dev, dtype, B, H, W, C1, C2, groups):
    """omg_groupnorm_mx8 == oracle quantiser applied to omg_groupnorm's 16-bit output, bit for bit (bytes and per-pixel scales)."""
    C = C1 + C2
    x1 = gen((B, H, W, C1), 11, dtype=dtype).to(dev)
    x2 = gen((B, H, W, C2), 12, scale=3.0, dtype=dtype).to(dev) if C2 else None
    gamma, beta = gen((C,), 13, dtype=dtype).to(dev), gen((C,), 14, dtype=dtype).to(dev)
    y = ops.groupnorm(x1, gamma, beta, groups, 1e-5, silu=True, x2=x2)
    got = ops.groupnorm_mx8(x1, gamma, beta, groups, 1e-5, silu=True, x2=x2)
    Cq = (C + 127) // 128 * 128                     # 320 -> 384, 960 -> 1024: pad channels must come out as zeros with scale byte 0
    yp = torch.zeros(B * H * W, Cq)
    yp[:, :C] = y.float().cpu().reshape(-1, C)
    q, packed, _ = mx8.quantize(yp)
    assert got.q.shape == (B, H, W, Cq) and got.scales.shape == (Cq // 128, B * H * W)
    assert torch.equal(got.q.cpu().reshape(-1, Cq), q), f"{(got.q.cpu().reshape(-1, Cq) != q).sum().item()} element bytes differ"
    assert torch.equal(got.scales.cpu(), packed)


def _conv_ref(dx, dw_packed, bias, gbias, res, out_scale):
    """dx (B,H,W,C) fp32, dw_packed (Cout, 9C) fp32 in (ky, kx, c) order -> NHWC fp32."""
    Cout, C = dw_packed.shape[0], dx.shape[-1]
    w = dw_packed.reshape(Cout, 3, 3, C).permute(0, 3, 1, 2)
    y = F.conv2d(dx.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    y = y + bias.float()
    if gbias is not None:
        y = y + gbias.float()[:, None, None, :]
    y = y * out_scale
    if res is not None:
        y = y + res.float()
    return y


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C,Cout", [(2, 16, 16, 128, 256), (3, 12, 20, 256, 136), (5, 8, 8, 128, 320), (1, 32, 32, 640, 640)])
def test_conv2d_mx8_matches_dequantised_fp32(dev, dtype, B, H, W, C, Cout):
    """The MX-fp8 implicit-GEMM convolution against F.conv2d (fp32, CPU) of the DEQUANTISED operands: exact products, fp32
    accumulation — tolerance is accumulation order + the 16-bit store.  Cases: whole tiles, ragged rows / columns with a map
    width that is no power of two, samples smaller than a tile (per-row group bias), K = 5760."""
    x = gen((B, H, W, C), 21, dtype=dtype)
    x[..., 7] *= 25.0                                  # an outlier channel
    x[0, 0, 0, :] = 0                                  # a pixel of zeros (scale byte 0)
    w = gen((Cout, 9 * C), 22, scale=(9 * C) ** -0.5, dtype=dtype)
    bias, gb, res = gen((Cout,), 23, dtype=dtype), gen((B, Cout), 24, dtype=dtype), gen((B, H, W, Cout), 25, dtype=dtype)
    M = B * H * W
    assert M % 4 == 0
    tx = ops.quant_mx8(x.reshape(M, C).to(dev))
    xm = ops.Mx8Map(tx.q.view(B, H, W, C), tx.scales, dtype)
    qx, _, ex = mx8.quantize(x.reshape(M, C).float())
    dx = mx8.dequantize(qx, ex).reshape(B, H, W, C)
    tw, dw = _mx(w, dev)
    tol = dict(rtol=2e-3, atol=3e-3) if dtype == torch.float16 else dict(rtol=1.6e-2, atol=2e-2)
    out = ops.conv2d_mx8(xm, tw, bias=bias.to(dev), group_bias=gb.to(dev))
    torch.testing.assert_close(out.float().cpu(), _conv_ref(dx, dw, bias, gb, None, 1.0), **tol)
    out = ops.conv2d_mx8(xm, tw, bias=bias.to(dev), residual=res.to(dev), out_scale=0.5)
    torch.testing.assert_close(out.float().cpu(), _conv_ref(dx, dw, bias, None, res, 0.5), **tol)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_resnet_block_in_mx8_mode_vs_fp16_mode(dev, dtype):
    """One ResnetBlock2D (concat input, shortcut conv, time-embedding bias) with conv1 / conv2 on the fp8 MFMA against the same
    block in 16 bits: relative rms difference at the fp8 quantisation level (e4m3: 3 mantissa bits -> ~2-3 % per operand)."""
    from omg_amd.unet import ResnetBlock2D, _Ctx
    torch.manual_seed(0)
    blk = ResnetBlock2D(384, 320, 512, 32, 1e-5, dtype, dev)      # conv2: 320 input channels -> padded to 384
    for n, p in blk.named_parameters():
        fan = p[0].numel() if p.dim() > 1 else 1
        p.data.copy_((torch.randn(p.shape) * (fan ** -0.5 if p.dim() > 1 else 0.1) + (1.0 if n.endswith("norm1.weight") or n.endswith("norm2.weight") else 0.0)).to(dtype))
    x, x2 = gen((2, 16, 16, 256), 31, dtype=dtype).to(dev), gen((2, 16, 16, 128), 32, dtype=dtype).to(dev)
    ctx = _Ctx(gen((2, 512), 33, dtype=dtype).to(dev), None, 2)
    y16 = blk(x, ctx, x2=x2).float().cpu()
    blk.conv1.mx8 = blk.conv2.mx8 = True
    assert blk.conv1.mx8_ok() and blk.conv2.mx8_ok() and not blk.conv_shortcut.mx8_ok()
    y8 = blk(x, ctx, x2=x2).float().cpu()
    rel = (y8 - y16).pow(2).mean().sqrt().item() / y16.pow(2).mean().sqrt().item()
    print(f"ResnetBlock2D {dtype}: MX-fp8 convolutions vs 16-bit, relative rms difference {rel:.2e}")
    assert 1e-4 < rel < 5e-2


def test_conv2d_mx8_at_bench_size_on_sampled_pixels(dev):
    """omg_conv2d_mx8 at two of the benchmark's launch shapes (8 requests per step: 64 samples of 32x32x1280 -> 1280, K = 11520;
    16 samples of 64x64x960 (padded to 1024) -> 640): ~1500 sampled output pixels incl. map corners / edges and tile boundaries
    against an fp32 CPU product of the dequantised 3x3 patches."""
    dtype = torch.float16
    for (B, H, W, C, Cout) in [(64, 32, 32, 1280, 1280), (16, 64, 64, 960, 640)]:
        g = torch.Generator(device=dev).manual_seed(7)
        x = torch.randn((B, H, W, C), generator=g, device=dev).to(dtype)
        gamma, beta = (torch.randn(C, generator=g, device=dev) * 0.2 + 1).to(dtype), (torch.randn(C, generator=g, device=dev) * 0.1).to(dtype)
        w = (torch.randn((Cout, 9, C), generator=g, device=dev) * (9 * C) ** -0.5).to(dtype)
        bias = torch.randn(Cout, generator=g, device=dev).to(dtype)
        gb = torch.randn((B, Cout), generator=g, device=dev).to(dtype)
        res = torch.randn((B, H, W, Cout), generator=g, device=dev).to(dtype)
        xm = ops.groupnorm_mx8(x, gamma, beta, 32, 1e-5, silu=True)
        Cq = xm.shape[-1]
        wp = torch.zeros((Cout, 9, Cq), dtype=dtype, device=dev)
        wp[:, :, :C] = w
        tw = ops.quant_mx8(wp.view(Cout, 9 * Cq))
        out = ops.conv2d_mx8(xm, tw, bias=bias, group_bias=gb, residual=res, out_scale=0.5)
        M = B * H * W
        ex = mx8.unpack_scales(xm.scales.cpu(), M)                                        # [M, Cq/32]
        dx = mx8.dequantize(xm.q.cpu().reshape(M, Cq), ex).reshape(B, H, W, Cq)
        dw = mx8.dequantize(tw.q.cpu(), mx8.unpack_scales(tw.scales.cpu(), Cout)).reshape(Cout, 9 * Cq)
        gi = torch.Generator().manual_seed(3)
        pix = torch.cat([torch.tensor([0, W - 1, (H - 1) * W, H * W - 1, H * W, 255, 256, M - 257, M - 256, M - 1]), torch.randint(0, M, (1500,), generator=gi)]).unique()
        b, y, xx = pix // (H * W), (pix % (H * W)) // W, pix % W
        dxp = F.pad(dx, (0, 0, 1, 1, 1, 1))                                               # zero ring = the convolution's padding
        patches = torch.stack([dxp[b, y + ky, xx + kx] for ky in range(3) for kx in range(3)], dim=1).reshape(len(pix), 9 * Cq)
        ref = (patches @ dw.T + bias.float().cpu() + gb.float().cpu()[b]) * 0.5 + res.float().cpu().reshape(M, Cout)[pix]
        err = (out.reshape(M, Cout)[pix.to(dev)].float().cpu() - ref).abs()
        assert (err <= 3e-3 + 2e-3 * ref.abs()).all(), (B, H, W, C, Cout, err.max().item())
        print(f"conv_mx8 {B}x{H}x{W}x{C}->{Cout}: {len(pix)} pixels, max |d| vs dequantised fp32 {err.max().item():.2e}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,slots", [(512, 512, 256, False), (300, 1024, 384, False), (2048, 2560, 640, True)])
def test_geglu_epilogue_mx8_output_equals_quantising_the_16bit_result(dev, dtype, M, N, K, slots):
    """omg_gemm_mx8 with c_scale (FeedForward: GEGLU feeding an MX-fp8 Linear): bytes and scale dwords must equal omg_quant_mx8
    of the 16-bit GEGLU output bit for bit — ragged rows, two column tiles, per-sample weight slots."""
    a = gen((M, K), 41, dtype=dtype).to(dev)
    nsl = 3 if slots else 1
    w = gen((nsl * N, K), 42, scale=K ** -0.5, dtype=dtype).to(dev)
    b = gen((N,), 43, dtype=dtype).to(dev)
    ta, tw = ops.quant_mx8(a), ops.quant_mx8(w)
    kw = dict(bias=b, act=L.ACT_GEGLU, out_dtype=dtype)
    if slots:
        kw.update(groups=8, w_group_adapter=torch.tensor([0, 0, 1, 2, 0, 1, 2, 2], dtype=torch.int32, device=dev), n_per_adapter=N)
    y16 = ops.gemm_mx8(ta, tw, **kw)
    two = ops.quant_mx8(y16)
    fused = ops.gemm_mx8(ta, tw, out_mx8=True, **kw)
    assert fused.q.shape == (M, N // 2) and fused.shape == (M, N // 2)
    assert torch.equal(fused.q, two.q), f"{(fused.q != two.q).sum().item()} bytes differ"
    assert torch.equal(fused.scales[:, :M], two.scales[:, :M])
