"""CPU tests (-m "not gpu"): pin the oracle against golden vectors produced by the reference's
own code (tests/golden/make_golden.py) and against the few externally known facts."""
import os

import numpy as np
import pytest
import torch

from oracle import controller as oc
from oracle import controlnet as ocn
from oracle import ip_adapter as oip
from oracle import pipeline as opipe
from oracle import schedulers as osched
from oracle import unet as ounet

GOLD = os.path.join(os.path.dirname(__file__), "golden")
P = "a man and a woman walking on the street"


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "controller_golden.npz"))


def make_cli_controller():
    return oc.AttentionReplaceOracle([P, P], 50, {"default_": 1.0}, 0.4, 32, 32, tokenizer=oc.WhitespaceTokenizer())


def test_T1_construction(gold):
    c = make_cli_controller()
    assert np.array_equal(c.mapper.numpy(), gold["t1_mapper"])
    assert np.array_equal(c.mapper.numpy()[0], np.eye(77, dtype=np.float32))
    assert np.array_equal(c.cross_replace_alpha.numpy(), gold["t1_alpha"])
    assert c.cross_replace_alpha.shape == (51, 1, 1, 1, 77) and bool((c.cross_replace_alpha == 1).all())
    assert tuple(gold["t1_num_self_replace"]) == c.num_self_replace == (0, 20)
    assert int(gold["t1_batch_size"]) == c.batch_size == 2


@pytest.mark.parametrize("name,is_cross,step", [("cross_q64", True, 0), ("self_q1024_s0", False, 0), ("self_q1056_s0", False, 0),
                                                  ("self_q1024_s19", False, 19), ("self_q1024_s20", False, 20)])
def test_T2_T3_T4_edit_matches_reference_bitwise(gold, name, is_cross, step):
    c = make_cli_controller()
    c.num_att_layers = 4
    c.cur_step = step
    x = torch.from_numpy(gold[name + "_in"].copy())
    y = c(x, is_cross, "down")
    assert y is x, "the controller edits in place and returns the same tensor"
    assert np.array_equal(y.numpy(), gold[name + "_out"])
    h = x.shape[0] // 4
    ref_in = gold[name + "_in"]
    assert np.array_equal(y.numpy()[: 3 * h], ref_in[: 3 * h])          # unc0, unc1, cond0 untouched
    replaced = np.array_equal(y.numpy()[3 * h:], ref_in[2 * h: 3 * h])  # cond1 := cond0 ?
    expect = is_cross or (x.shape[1] <= 32 * 32 and step < 20)
    assert replaced == expect


def test_T5_counters(gold):
    c = make_cli_controller()
    c.num_att_layers = 140
    t = torch.softmax(torch.randn(4, 2, 77), -1)
    for _ in range(140):
        c(t, True, "down")
    assert [c.cur_step, c.cur_att_layer] == list(gold["t5_counters"]) == [1, 0]


def test_T6_mappers(gold):
    m = oc.replacement_mapper(["a man on the street", "a dog on the street"], oc.WhitespaceTokenizer())
    assert np.array_equal(m.numpy(), gold["t6_mapper_swap"])
    assert int(gold["t6_raises"]) == 1
    with pytest.raises(ValueError):
        oc.replacement_mapper(["a man", "a man walking"], oc.WhitespaceTokenizer())
    m2 = oc.replacement_mapper(["a man on the road", "a woman on the road"], oc.PieceTokenizer())
    assert np.array_equal(m2.numpy(), gold["mapper_pieces"])
    assert not np.array_equal(m2.numpy()[0], np.eye(77, dtype=np.float32))


@pytest.mark.parametrize("step", [0, 3, 7])
@pytest.mark.parametrize("kind", ["cross", "self"])
def test_general_mapper_path_matches_reference(gold, step, kind):
    c = oc.AttentionReplaceOracle(["a man on the road", "a woman on the road"], 10, {"default_": 0.6, "road": (0.2, 0.9)},
                                  (0.1, 0.5), 4, 4, tokenizer=oc.PieceTokenizer())
    assert np.array_equal(c.cross_replace_alpha.numpy(), gold["gen_alpha"])
    assert np.array_equal(c.mapper.numpy(), gold["gen_mapper"])
    assert tuple(gold["gen_num_self_replace"]) == c.num_self_replace
    c.num_att_layers = 2
    c.cur_step = step
    key = f"gen_s{step}_{kind}"
    y = c(torch.from_numpy(gold[key + "_in"].copy()), kind == "cross", "mid")
    np.testing.assert_allclose(y.numpy(), gold[key + "_out"], rtol=0, atol=1e-7)


def test_ip_adapter_processor_matches_reference():
    g = np.load(os.path.join(GOLD, "ip_adapter_golden.npz"))
    T = lambda k: torch.from_numpy(g[k])
    C, ctx, heads, ntok = (int(v) for v in g["meta"])
    y = oip.ip_cross_attention(T("hidden_states"), T("encoder_hidden_states"), T("attn.to_q.weight"), T("attn.to_k.weight"),
                               T("attn.to_v.weight"), T("attn.to_out.0.weight"), T("attn.to_out.0.bias"), T("to_k_ip"),
                               T("to_v_ip"), heads, 0.8, ntok)
    np.testing.assert_allclose(y.numpy(), g["out_cross"], rtol=1e-5, atol=2e-6)
    ys = oip.self_attention(T("hidden_states"), T("self_attn.to_q.weight"), T("self_attn.to_k.weight"), T("self_attn.to_v.weight"),
                            T("self_attn.to_out.0.weight"), T("self_attn.to_out.0.bias"), heads)
    np.testing.assert_allclose(ys.numpy(), g["out_self"], rtol=1e-5, atol=2e-6)


def test_unet_topology_parameter_count_is_sdxl():
    n = 0
    for shp in ounet.param_shapes(ounet.UNetConfig.sdxl()).values():
        k = 1
        for d in shp:
            k *= d
        n += k
    assert n == 2_567_463_684          # published SDXL-base UNet size
    assert ounet.count_attention_layers(ounet.UNetConfig.sdxl()) == 70   # lora_pipeline.py:152 -> num_att_layers = 140


def test_tiny_unet_runs_and_controller_counts_one_step():
    cfg = ounet.UNetConfig.tiny()
    sd = ounet.init_state_dict(cfg, seed=0)
    B, L = 4, cfg.sample_size
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 4, L, L, generator=g)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    te = torch.randn(B, 64, generator=g)
    tid = torch.tensor([[L * 8, L * 8, 0, 0, L * 8, L * 8]] * B, dtype=torch.float32)
    c = oc.AttentionReplaceOracle([P, P], 50, {"default_": 1.0}, 0.4, L // 4, L // 4)
    c.num_att_layers = 2 * ounet.count_attention_layers(cfg)
    y = ounet.unet_forward(sd, cfg, x, 981, ctx, te, tid, attn_fn=oc.reference_attn_fn(c))
    assert y.shape == x.shape and torch.isfinite(y).all()
    assert (c.cur_step, c.cur_att_layer) == (1, 0)
    y_plain = ounet.unet_forward(sd, cfg, x, 981, ctx, te, tid)
    assert torch.allclose(y[:3], y_plain[:3], atol=1e-5)     # unc0/unc1/cond0 never depend on the controller
    assert not torch.allclose(y[3], y_plain[3], atol=1e-4)   # cond1 does (x identical, but ctx differs per sample)


def test_fusion_overlap_sums_and_none_masks():
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(4, 4, 16, 16, generator=g)
    r1, r2 = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 4, 16, 16, generator=g)
    m1 = torch.zeros(128, 128); m1[32:, 8:60] = 1
    m2 = torch.zeros(128, 128); m2[32:, 56:120] = 1
    out = opipe.fuse_noise(noise, [r1, r2], [m1, m2])
    a, b = m1[::8, ::8] == 1, m2[::8, ::8] == 1      # nearest resize = m[8y, 8x]
    assert torch.equal(out[0], noise[0]) and torch.equal(out[2], noise[2])
    for slot, j in ((1, 0), (3, 1)):
        exp = torch.where(a | b, torch.zeros(()), noise[slot]) + a * r1[j] + b * r2[j]
        assert torch.allclose(out[slot], exp, atol=1e-6)
    assert (a & b).any()
    out2 = opipe.fuse_noise(noise, [r1, None], [m1, None])
    exp = torch.where(a, r1[1], noise[3])
    assert torch.allclose(out2[3], exp, atol=1e-6)
    out3 = opipe.fuse_noise(noise, [None, None], [None, None])
    assert torch.equal(out3, noise)


def test_schedulers_basic_properties():
    d = osched.DDIM(50)
    assert list(d.timesteps[:3]) == [981, 961, 941] and d.timesteps[-1] == 1
    # DDIM with the true eps is exact: x_t = sqrt(a) x0 + sqrt(1-a) e  ->  step gives x_{t-1} with same x0,e
    x0, e = np.float64(0.7), np.float64(-1.3)
    a_t = d.ac[981]; a_p = d.ac[961]
    xt = a_t ** 0.5 * x0 + (1 - a_t) ** 0.5 * e
    assert abs(d.step(e, 0, xt) - (a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * e)) < 1e-12
    eu = osched.EulerDiscrete(50)
    assert np.all(np.diff(eu.sigmas) < 0) and eu.sigmas[-1] == 0
    assert abs(eu.init_noise_sigma - (eu.sigmas[0] ** 2 + 1) ** 0.5) < 1e-12
    # externally known constants of the SD / SDXL noise schedule (scaled-linear betas 0.00085 .. 0.012 over 1000 steps): the sigma range
    # every k-diffusion-style sampler quotes for these checkpoints, sigma_min 0.0292 and sigma_max 14.6146 (recalled from public
    # documentation, not from a file in /root/reference — an anchor, not a pin), and alphas_cumprod[999] = 0.00466
    ac = osched.alphas_cumprod()
    assert abs(((1 - ac[999]) / ac[999]) ** 0.5 - 14.6146) < 1e-3 and abs(((1 - ac[0]) / ac[0]) ** 0.5 - 0.0292) < 1e-4
    assert abs(ac[999] - 0.00466) < 1e-5
    from omg_amd.schedulers import EulerDiscreteScheduler
    mine = EulerDiscreteScheduler(); mine.set_timesteps(50, device="cpu")
    assert abs(float(mine.alphas_cumprod[999]) - ac[999]) < 1e-12          # the product's own table is the same schedule


def test_controlnet_topology_parameter_count():
    n = 0
    for shp in ocn.param_shapes(ounet.UNetConfig.sdxl()).values():
        k = 1
        for d in shp:
            k *= d
        n += k
    assert n == 1_251_014_160        # SDXL ControlNet: encoder copy of the UNet + conditioning embedding + 10 zero convs
    assert len([k for k in ocn.param_shapes(ounet.UNetConfig.sdxl()) if k.startswith("controlnet_down_blocks.") and k.endswith(".weight")]) == 9


# ------------------------------------------------------------------ VAE decoder oracle (row N1) — anchors available without a GPU
def test_vae_oracle_topology_and_key_layout():
    import math
    from oracle import vae as ov
    from omg_amd.vae import AutoencoderKLDecoder, VaeConfig
    shapes = ov.param_shapes(ov.VaeConfig.sdxl())
    assert sum(math.prod(s) for k, s in shapes.items() if k.startswith("decoder.")) == 49_490_179      # SDXL VAE decoder
    assert sum(math.prod(s) for k, s in shapes.items() if k.startswith("post_quant_conv.")) == 20
    prod = AutoencoderKLDecoder(VaeConfig.sdxl(), dtype=torch.float16, device="meta")
    assert {k: tuple(v.shape) for k, v in prod.state_dict().items()} == shapes                        # diffusers' key layout, both sides
    cfg = ov.VaeConfig.tiny()
    sd = ov.init_state_dict(cfg, seed=0)
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    taps = {}
    img = ov.decode(sd, cfg, z, taps)
    assert img.shape == (2, 3, 16, 16) and torch.isfinite(img).all() and set(taps) == {"mid", "up0", "up1"}
    # batch-independent (torch-CPU picks different conv algorithms per batch size, so to rounding, not bitwise)
    torch.testing.assert_close(ov.decode(sd, cfg, z[:1]), img[:1], rtol=1e-5, atol=1e-5)
    assert float(ov.postprocess(img).min()) >= 0.0 and float(ov.postprocess(img).max()) <= 1.0


# ------------------------------------------------------------------ Resampler (InstantID image_proj_model) — pinned by the reference's own class
def test_resampler_oracle_matches_reference_golden():
    from oracle import resampler as orr
    g = np.load(os.path.join(GOLD, "resampler_golden.npz"))
    dim, depth, dim_head, heads, nq, emb, outd, mult = (int(v) for v in g["cfg"])
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    assert {k: tuple(v.shape) for k, v in sd.items()} == orr.param_shapes(dim, depth, dim_head, heads, nq, emb, outd, mult)
    y = orr.resampler_forward(sd, torch.from_numpy(g["x"]), heads)
    torch.testing.assert_close(y, torch.from_numpy(g["y"]), rtol=1e-5, atol=1e-5)
    # InstantID's configuration (instantid_single_pieline.py:165-174): 16 tokens of 2048 from one 512-d embedding
    full = orr.param_shapes(1280, 4, 64, 20, 16, 512, 2048, 4)
    assert full["latents"] == (1, 16, 1280) and full["proj_out.weight"] == (2048, 1280) and full["layers.3.1.3.weight"] == (1280, 5120)


def test_storage_precision_emulation_of_the_oracle_unet():
    """oracle/precision.py (round 4): inside ``rounding(torch.float16)`` every op's output is rounded as the reference's fp16 eager run
    rounds it; outside, the oracle is bit for bit the fp32 function it was.  Facts checked: (i) no state leaks out of the context;
    (ii) the emulated output is fp16-representable and differs from fp32 by ~1e-3 of the output rms (fp16 has 11 significant bits, ~60
    chained layers) — not by 0 (the hook would be dead) and not by 1e-1 (it would be broken); (iii) bf16 is coarser than fp16;
    (iv) with the controller's attention sequence the rounding reaches the probabilities the controller edits."""
    from oracle import precision as oprec
    cfg = ounet.UNetConfig.tiny()
    sd = ounet.init_state_dict(cfg, seed=0, dtype=torch.float16)
    B, L = 4, cfg.sample_size
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 4, L, L, generator=g)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g).half().float()
    te = torch.randn(B, 64, generator=g).half().float()
    tid = torch.tensor([[L * 8, L * 8, 0, 0, L * 8, L * 8]] * B, dtype=torch.float32)
    names = ounet.lora_target_names(cfg)
    _, lora = ounet.make_lora(cfg, names, rank=4, seed=3, scale=0.8, dtype=torch.float16)

    def run(attn=None):
        kw = {} if attn is None else {"attn_fn": attn}
        return ounet.unet_forward(sd, cfg, x, 981, ctx, te, tid, lora=lora, **kw)

    y32 = run()
    with oprec.rounding(torch.float16):
        assert oprec.active() == torch.float16
        y16 = run()
    with oprec.rounding(torch.bfloat16):
        yb = run()
    assert oprec.active() is None and torch.equal(run(), y32)
    assert torch.equal(y16, y16.half().float())
    rms = y32.pow(2).mean().sqrt()
    e16, eb = ((y16 - y32).pow(2).mean().sqrt() / rms).item(), ((yb - y32).pow(2).mean().sqrt() / rms).item()
    assert 1e-4 < e16 < 1e-2 and e16 < eb < 1e-1, (e16, eb)
    c = oc.AttentionReplaceOracle([P, P], 50, {"default_": 1.0}, 0.4, L // 4, L // 4)
    c.num_att_layers = 2 * ounet.count_attention_layers(cfg)
    yc32 = run(oc.reference_attn_fn(c))
    c.reset()
    with oprec.rounding(torch.float16):
        yc16 = run(oc.reference_attn_fn(c))
    ec = ((yc16 - yc32).pow(2).mean().sqrt() / yc32.pow(2).mean().sqrt()).item()
    assert 1e-4 < ec < 1e-2, ec
