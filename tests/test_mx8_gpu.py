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
    for mode in ("fp16", "mx8"):
        unet.set_linear_precision(mode)
        unet.set_lora_state(state)
        out[mode] = unet(x.to(dev), 500.0, encoder_hidden_states=ctx.to(dev).to(dtype),
                         added_cond_kwargs={"text_embeds": te.to(dev).to(dtype), "time_ids": tid.to(dev)})[0].float().cpu()
        unet.set_lora_state(None)
    unet.set_linear_precision("fp16")
    rms = ref.pow(2).mean().sqrt().item()
    e16 = (out["fp16"] - ref).abs().max().item() / rms
    e8 = (out["mx8"] - ref).abs().max().item() / rms
    e8r = (out["mx8"] - ref).pow(2).mean().sqrt().item() / rms
    print(f"UNet forward (lora={lora}) vs fp32 oracle, max |d| / rms: fp16 path {e16:.2e}, MX-fp8 Linear path {e8:.2e} (rms error {e8r:.2e}); "
          f"fp8 vs fp16 path {(out['mx8'] - out['fp16']).abs().max().item() / rms:.2e}")
    assert e16 < 3e-2
    assert e8 < 0.5 and e8r < 0.1        # measured: see profiles/r02_mx8_precision.txt
    assert not torch.equal(out["mx8"], out["fp16"])
