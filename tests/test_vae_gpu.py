"""Row N1: AutoencoderKL.decode on the HIP kernels vs the fp32 oracle (oracle/vae.py).
Tolerances: the decoder is ~30 convolution / normalisation layers deep with 16-bit storage between them; the checks are
on the error relative to the output's rms (fp16: 1.5e-2 rms, 6e-2 max; bf16: 6e-2 rms, 2.5e-1 max)."""
import pytest
import torch
import torch.nn.functional as F

from omg_amd import ops
from omg_amd.vae import AutoencoderKLDecoder, VaeConfig
from oracle import vae as ov

pytestmark = pytest.mark.gpu
TOL = {torch.float16: (1.5e-2, 6e-2), torch.bfloat16: (6e-2, 2.5e-1)}


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _build(cfg_o, cfg_p, dtype, dev, seed=0):
    sd = ov.init_state_dict(cfg_o, seed=seed)
    vae = AutoencoderKLDecoder(cfg_p, dtype=dtype, device=dev)
    vae.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    # the oracle sees the same 16-bit rounded weights (post_quant_conv stays fp32 on both sides)
    sd_r = {k: (v if k.startswith("post_quant_conv") else v.to(dtype).float()) for k, v in sd.items()}
    return sd_r, vae


def _check(out, ref, dtype):
    rms_tol, max_tol = TOL[dtype]
    ref = ref.float()
    diff = out.float().cpu() - ref
    rms = ref.pow(2).mean().sqrt()
    assert torch.isfinite(out).all()
    assert diff.pow(2).mean().sqrt() / rms < rms_tol, f"rms error {diff.pow(2).mean().sqrt() / rms:.3e}"
    assert diff.abs().max() / rms < max_tol, f"max error {diff.abs().max() / rms:.3e}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_softmax_rows_and_channel_mix(dev, dtype):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(70, 1000, generator=g) * 3).to(dtype)
    big = torch.zeros(70, 1024, dtype=dtype)
    big[:, :1000] = x
    xd = big.to(dev)
    ops.softmax_rows_(xd[:, :1000], 0.37)
    ref = torch.softmax(x.float() * 0.37, dim=-1)
    torch.testing.assert_close(xd[:, :1000].float().cpu(), ref, rtol=2e-2 if dtype == torch.bfloat16 else 4e-3, atol=1e-5)
    assert torch.equal(xd[:, 1000:].cpu(), torch.zeros(70, 24, dtype=dtype))           # padding columns untouched
    z = torch.randn(3, 4, 9, 7, generator=g)
    w, b = torch.randn(4, 4, 1, 1, generator=g), torch.randn(4, generator=g)
    y = ops.channel_mix(z.to(dev), w.reshape(4, 4).contiguous().to(dev), b.to(dev))
    torch.testing.assert_close(y.cpu(), F.conv2d(z, w, b), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_decode_tiny_matches_oracle(dev, dtype):
    cfg_o = ov.VaeConfig.tiny()
    sd, vae = _build(cfg_o, VaeConfig.tiny(), dtype, dev)
    z = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(1))
    taps = {}
    ref = ov.decode(sd, cfg_o, z, taps)
    out = vae.decode(z.to(dev))
    assert out.shape == (2, 3, 32, 32) and out.dtype == torch.float32
    _check(out, ref, dtype)
    # the reference's tail: divide by the scaling factor, decode, denormalise (lora_pipeline.py:650-661)
    img = vae.decode_latents(z.to(dev) * cfg_o.scaling_factor)
    ref_img = ov.postprocess(ref)
    assert (img.cpu() - ref_img).abs().max() < (0.06 if dtype == torch.float16 else 0.2)


def test_decode_sdxl_width_small_latent(dev):
    """Full SDXL channel widths (512/512/256/128, 3 resnets per block, 512-dim single-head attention) on a 16x16 latent."""
    dtype = torch.float16
    cfg_o = ov.VaeConfig.sdxl()
    sd, vae = _build(cfg_o, VaeConfig.sdxl(), dtype, dev, seed=3)
    z = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(2))
    ref = ov.decode(sd, cfg_o, z)
    out = vae.decode(z.to(dev))
    assert out.shape == (1, 3, 128, 128)
    _check(out, ref, dtype)


def test_decoder_refuses_cpu_tensors(dev):
    vae = AutoencoderKLDecoder(VaeConfig.tiny(), dtype=torch.float16, device=dev).init_synthetic_(0)
    with pytest.raises(Exception):
        vae.decode(torch.zeros(1, 4, 8, 8))


def test_pipeline_decodes_through_the_vae(dev):
    """output_type != 'latent' runs the reference's tail (lora_pipeline.py:635-661) on the decoder attached as vae_decode."""
    import contextlib, io
    from omg_amd import controller as pc
    from omg_amd.pipeline import LoraMultiConceptPipeline, revise_regionally_controlnet_forward
    from omg_amd.schedulers import make_scheduler
    from omg_amd.synthetic import c2_inputs, c2_masks, make_concept_models
    from omg_amd.unet import UNet2DConditionModel, UNetConfig
    dtype = torch.float16
    cfg = UNetConfig.tiny()
    unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev).init_synthetic_(0)
    HW = cfg.sample_size * 8
    P = "a man and a woman"
    ctl = pc.AttentionReplace([P, P], 4, {"default_": 1.0}, 0.4, HW // 32, HW // 32, device=dev, dtype=dtype)
    with contextlib.redirect_stdout(io.StringIO()):
        revise_regionally_controlnet_forward(unet, ctl)
    concept = make_concept_models(unet, n_concepts=2, rank=8)
    vae = AutoencoderKLDecoder(VaeConfig.tiny(), dtype=dtype, device=dev).init_synthetic_(0)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"), vae_decode=vae.decode_latents)
    kw = dict(num_inference_steps=4, height=HW, width=HW, guidance_scale=7.5, cross_attention_kwargs={"scale": 0.8}, controller=ctl,
              concept_models=concept, stage=2, region_masks=c2_masks(HW, HW, device=dev), lora_list=["concept0", "concept1"],
              styleL=False, **c2_inputs(unet, 0, height=HW, width=HW))
    lat = pipe(output_type="latent", **kw).images
    ctl.reset()
    img = pipe(output_type="pt", **kw).images
    up = 2 ** (len(vae.config.block_out_channels) - 1)          # 2 for the tiny decoder, 8 for SDXL's
    assert img.shape == (lat.shape[0], 3, lat.shape[2] * up, lat.shape[3] * up) and img.dtype == torch.float32
    assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    assert torch.equal(img, vae.decode_latents(lat))


# ------------------------------------------------------------------ round 2: the reference's fp32 ("upcast") decode
def test_fp32_kernels_match_torch_fp32(dev):
    """omg_conv2d_f32 (f32-input MFMA), fp32 GroupNorm(+SiLU), fp32 conv_out, cast: vs torch fp32 on the CPU.  The MFMA's products are
    exact fp32 and accumulate in fp32 in a different order than torch's: tolerance = fp32 summation noise, not 16-bit rounding."""
    g = torch.Generator().manual_seed(0)
    for (B, H, Cin, Cout, k, ups) in [(2, 16, 128, 128, 3, False), (1, 24, 256, 128, 3, True), (2, 20, 64, 96, 1, False), (1, 33, 32, 260, 3, False)]:
        x = torch.randn(B, H, H, Cin, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) * (k * k * Cin) ** -0.5
        b = torch.randn(Cout, generator=g)
        Ho = 2 * H if ups else H
        res = torch.randn(B, Ho, Ho, Cout, generator=g)
        y = ops.conv2d_f32(x.to(dev), ops.pack_conv_weight(w.to(dev)), k, upsample=ups, bias=b.to(dev), residual=res.to(dev))
        xin = x.permute(0, 3, 1, 2)
        if ups:
            xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
        ref = F.conv2d(xin.double(), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1) + res.double()
        err = (y.cpu().double() - ref).abs().max().item()
        assert err < 2e-5, (B, H, Cin, Cout, k, ups, err)
    x = torch.randn(2, 40, 40, 128, generator=g) * 3 + 1
    ga, be = torch.randn(128, generator=g), torch.randn(128, generator=g)
    y = ops.groupnorm(x.to(dev), ga.to(dev), be.to(dev), 32, 1e-6, silu=True)
    ref = F.silu(F.group_norm(x.permute(0, 3, 1, 2).double(), 32, ga.double(), be.double(), 1e-6)).permute(0, 2, 3, 1)
    assert y.dtype == torch.float32 and (y.cpu().double() - ref).abs().max() < 2e-5
    w = torch.randn(3, 128, 3, 3, generator=g) * 0.03
    b3 = torch.randn(3, generator=g)
    yo = ops.conv_out(x.to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), b3.to(dev))
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b3.double(), padding=1)
    assert (yo.cpu().double() - ref).abs().max() < 5e-5
    h = (torch.randn(4, 8, 8, 64, generator=g)).half()
    assert torch.equal(ops.cast_f32(h.to(dev)).cpu(), h.float())


def test_fp32_conv_addresses_beyond_32_bits_and_tiles_across_images(dev):
    """conv_f32_kernel's descriptor addressing (round 6): 32-bit lane offsets from the first image a 128-row tile touches.  (a) images smaller than a
    tile — one tile spans up to FBM / (H W) + 1 images, with and without the fused 2x upsample; (b) an input of 5.4 GB (5 x 1024 x 1024 x 256 fp32:
    the shape of the VAE's last up block at batch 5): the descriptor of a tile in the last image starts 4.3 GB into the tensor and its range field
    exceeds 2^31.  (b) is checked on sampled output pixels — corners, image borders, the last pixel of the last image — against a float64 dot product."""
    g = torch.Generator().manual_seed(1)
    for (B, H, Cin, Cout, k, ups) in [(5, 8, 32, 64, 3, False), (3, 4, 64, 128, 3, True), (7, 5, 32, 32, 1, False), (9, 3, 32, 96, 3, False)]:
        x = torch.randn(B, H, H, Cin, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) * (k * k * Cin) ** -0.5
        b = torch.randn(Cout, generator=g)
        y = ops.conv2d_f32(x.to(dev), ops.pack_conv_weight(w.to(dev)), k, upsample=ups, bias=b.to(dev))
        xin = x.permute(0, 3, 1, 2)
        if ups:
            xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
        ref = F.conv2d(xin.double(), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
        err = (y.cpu().double() - ref).abs().max().item()
        assert err < 2e-5, (B, H, Cin, Cout, k, ups, err)
    B, H, Cin, Cout = 5, 1024, 256, 128
    gd = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(B, H, H, Cin, generator=gd, device=dev)
    assert x.numel() * 4 > 4 * 2 ** 30                      # beyond any 32-bit byte offset
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5
    y = ops.conv2d_f32(x, ops.pack_conv_weight(w.to(dev)), 3)
    pts = [(0, 0, 0), (0, H - 1, H - 1), (1, 0, 0), (2, 511, 513), (3, H - 1, 0), (4, 0, H - 1), (4, 777, 3), (4, H - 1, H - 1), (4, H - 2, H - 2)]
    pts += [(int(torch.randint(0, B, (1,), generator=g)), int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, H, (1,), generator=g))) for _ in range(24)]
    wd = w.double()
    for (bi, yy, xx) in pts:
        acc = torch.zeros(Cout, dtype=torch.float64)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                iy, ix = yy + dy, xx + dx
                if 0 <= iy < H and 0 <= ix < H:
                    acc += wd[:, :, dy + 1, dx + 1] @ x[bi, iy, ix].cpu().double()
        err = (y[bi, yy, xx].cpu().double() - acc).abs().max().item()
        assert err < 2e-5, (bi, yy, xx, err)
    del x, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("cfg_name", ["tiny", "sdxl"])
def test_upcast_decode_matches_oracle(dev, cfg_name):
    """upcast=True = the reference's decode (lora_pipeline.py:639-652): post_quant_conv / conv_in / mid block in fp16, up blocks,
    conv_norm_out and conv_out in fp32 with fp32 weights.  The oracle gets fp16-rounded weights for the fp16 part only."""
    cfg_o = getattr(ov.VaeConfig, cfg_name)()
    cfg_p = getattr(VaeConfig, cfg_name)()
    sd = ov.init_state_dict(cfg_o, seed=4)
    vae = AutoencoderKLDecoder(cfg_p, dtype=torch.float16, device=dev, upcast=True)
    vae.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    assert vae.decoder.up_blocks[0].resnets[0].conv1.weight.dtype == torch.float32 and vae.decoder.mid_block.resnets[0].conv1.weight.dtype == torch.float16
    sd_r = {k: (v.half().float() if k.startswith(("decoder.conv_in", "decoder.mid_block")) else v) for k, v in sd.items()}
    z = torch.randn(2 if cfg_name == "tiny" else 1, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    ref = ov.decode(sd_r, cfg_o, z).float()
    out = vae.decode(z.to(dev)).float().cpu()
    rms = ref.pow(2).mean().sqrt()
    e_rms, e_max = ((out - ref).pow(2).mean().sqrt() / rms).item(), ((out - ref).abs().max() / rms).item()
    lo = AutoencoderKLDecoder(cfg_p, dtype=torch.float16, device=dev)
    lo.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    out16 = lo.decode(z.to(dev)).float().cpu()
    e16 = ((out16 - ref).pow(2).mean().sqrt() / rms).item()
    print(f"VAE decode {cfg_name}: upcast (fp16 mid, fp32 up blocks) rms err {e_rms:.2e} max {e_max:.2e}; all-fp16 rms err {e16:.2e}")
    assert e_rms < 5e-3 and e_max < 3e-2
    assert e_rms < e16
