"""Row N3: checkpoint / LoRA file loaders — host logic, CPU only."""
import os

import pytest
import torch

from omg_amd import loaders
from omg_amd.lora import LoraAdapter, lora_target_names
from omg_amd.unet import UNet2DConditionModel, UNetConfig


@pytest.fixture(scope="module")
def unet():
    return UNet2DConditionModel(UNetConfig.tiny(), dtype=torch.float16, device="cpu")


def _adapter(unet, rank=8, alpha=None, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = {}
    for key in lora_target_names(unet):
        lin = unet.get_submodule(key)
        w[key] = (torch.randn(rank, lin.in_features, generator=g), torch.randn(lin.out_features, rank, generator=g))
    return LoraAdapter("c0", w, alpha=alpha)


def _delta(ad, key):
    a, b = ad.weights[key]
    return (ad.alpha / ad.rank) * (b.float() @ a.float())


@pytest.mark.parametrize("style", ["peft", "diffusers", "kohya"])
def test_lora_round_trip_all_key_styles(unet, style, tmp_path):
    src = _adapter(unet, rank=8, alpha=4.0 if style == "kohya" else None)
    sd = loaders.lora_state_dict(src, style)
    sd["text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_A.weight"] = torch.zeros(8, 4)   # text-encoder half: parsed
    sd["text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_B.weight"] = torch.zeros(4, 8)
    from safetensors.torch import save_file
    d = tmp_path / "my_concept"
    os.makedirs(d)
    save_file(sd, str(d / "pytorch_lora_weights.safetensors"))
    got = loaders.load_lora_adapter(unet, str(d))                   # directory + default weight name, as the reference calls it
    assert got.name == "my_concept" and got.rank == 8
    assert set(got.text_encoder[1]) == {"text_model.encoder.layers.0.self_attn.q_proj"} and not got.skipped_keys
    assert set(got.weights) == set(src.weights)
    for key in list(src.weights)[::7]:
        torch.testing.assert_close(_delta(got, key), _delta(src, key), rtol=1e-6, atol=1e-6)


def test_lora_old_attention_processor_keys_and_adapter_segment(unet):
    src = _adapter(unet, rank=4)
    sd = {}
    for mod, (a, b) in src.weights.items():
        if ".attn" in mod and mod.rsplit(".", 1)[-1] in ("to_q", "to_k", "to_v"):
            attn, proj = mod.rsplit(".", 1)
            sd[f"unet.{attn}.processor.{proj}_lora.down.weight"], sd[f"unet.{attn}.processor.{proj}_lora.up.weight"] = a, b
        elif mod.endswith(".to_out.0"):
            attn = mod[: -len(".to_out.0")]
            sd[f"unet.{attn}.processor.to_out_lora.down.weight"], sd[f"unet.{attn}.processor.to_out_lora.up.weight"] = a, b
        else:
            sd[f"{mod}.lora_A.default.weight"], sd[f"{mod}.lora_B.default.weight"] = a, b        # PEFT with an adapter-name segment
    got, skipped = loaders.parse_lora_state_dict(sd, loaders.linear_module_paths(unet), "x")
    assert not skipped and set(got.weights) == set(src.weights)
    k = next(iter(src.weights))
    assert torch.equal(got.weights[k][0], src.weights[k][0].float())


def test_lora_errors(unet):
    paths = loaders.linear_module_paths(unet)
    k = lora_target_names(unet)[0]
    a, b = torch.zeros(4, unet.get_submodule(k).in_features), torch.zeros(unet.get_submodule(k).out_features, 4)
    with pytest.raises(loaders.LoaderError, match="no UNet LoRA entries"):
        loaders.parse_lora_state_dict({"something.else": a}, paths)
    with pytest.raises(loaders.LoaderError, match="incomplete"):
        loaders.parse_lora_state_dict({f"unet.{k}.lora_A.weight": a}, paths)
    with pytest.raises(loaders.LoaderError, match="not a Linear"):
        loaders.parse_lora_state_dict({"unet.conv_in.lora_A.weight": a, "unet.conv_in.lora_B.weight": b}, paths)
    k2 = lora_target_names(unet)[1]
    a2, b2 = torch.zeros(8, unet.get_submodule(k2).in_features), torch.zeros(unet.get_submodule(k2).out_features, 8)
    mixed, _ = loaders.parse_lora_state_dict({f"unet.{k}.lora_A.weight": a, f"unet.{k}.lora_B.weight": b,      # round 2: accepted
                                              f"unet.{k2}.lora_A.weight": a2, f"unet.{k2}.lora_B.weight": b2}, paths)
    assert mixed.rank == max(a.shape[0], 8)


def test_model_checkpoint_round_trip(unet, tmp_path):
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(1)
    sd = {k: torch.randn(v.shape, generator=g).to(v.dtype) for k, v in unet.state_dict().items()}
    path = str(tmp_path / "diffusion_pytorch_model.fp16.safetensors")
    save_file(sd, path)
    other = UNet2DConditionModel(UNetConfig.tiny(), dtype=torch.float16, device="cpu")
    assert loaders.load_model_weights(other, path) == []
    for k, v in other.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # a whole-pipeline file with a "unet." namespace and a stray tensor
    pref = {f"unet.{k}": v for k, v in sd.items()}
    pref["vae.decoder.conv_in.weight"] = torch.zeros(1)
    assert loaders.load_model_weights(other, pref, prefix="unet.") == []
    bad = dict(sd)
    bad.pop("conv_in.bias")
    with pytest.raises(loaders.LoaderError, match="1 missing"):
        loaders.load_model_weights(other, bad)
    bad = dict(sd)
    bad["conv_in.bias"] = torch.zeros(3)
    with pytest.raises(loaders.LoaderError, match="shape mismatch"):
        loaders.load_model_weights(other, bad)


def test_decoder_only_vae_loads_from_a_full_vae_file():
    from omg_amd.vae import AutoencoderKLDecoder, VaeConfig
    vae = AutoencoderKLDecoder(VaeConfig.tiny(), dtype=torch.float16, device="cpu")
    g = torch.Generator().manual_seed(2)
    sd = {k: torch.randn(v.shape, generator=g).to(v.dtype) for k, v in vae.state_dict().items()}
    full = dict(sd)
    full["encoder.conv_in.weight"] = torch.zeros(8, 3, 3, 3)
    full["quant_conv.weight"] = torch.zeros(8, 8, 1, 1)
    with pytest.raises(loaders.LoaderError, match="unexpected"):
        loaders.load_model_weights(vae, full)
    assert sorted(loaders.load_model_weights(vae, full, allow_extra=True)) == ["encoder.conv_in.weight", "quant_conv.weight"]
    assert all(torch.equal(v, sd[k]) for k, v in vae.state_dict().items())
    full.pop("decoder.conv_out.bias")
    with pytest.raises(loaders.LoaderError, match="1 missing"):
        loaders.load_model_weights(vae, full, allow_extra=True)


# ------------------------------------------------------------------ round 2: SGM names, text-encoder halves, mixed ranks
def _sgm_flat(mod: str) -> str:
    """diffusers module path -> kohya-ss SDXL (SGM block) flat name, written independently of the loader's inverse."""
    import re
    m = re.match(r"^down_blocks\.(\d+)\.(attentions|resnets)\.(\d+)\.(.*)$", mod)
    if m:
        b, kind, l, rest = int(m[1]), m[2], int(m[3]), m[4]
        head = f"input_blocks.{1 + 3 * b + l}.{1 if kind == 'attentions' else 0}."
    else:
        m = re.match(r"^up_blocks\.(\d+)\.(attentions|resnets)\.(\d+)\.(.*)$", mod)
        if m:
            b, kind, l, rest = int(m[1]), m[2], int(m[3]), m[4]
            head = f"output_blocks.{3 * b + l}.{1 if kind == 'attentions' else 0}."
        else:
            m = re.match(r"^mid_block\.(attentions|resnets)\.(\d+)\.(.*)$", mod)
            kind, l, rest = m[1], int(m[2]), m[3]
            head = f"middle_block.{1 if kind == 'attentions' else 2 * l}."
    rest = rest.replace("time_emb_proj", "emb_layers.1")
    return (head + rest).replace(".", "_")


def test_kohya_sgm_block_names_map_to_diffusers_modules(unet):
    """The reference's shipped concept files (kohya-ss SDXL LoRAs: chris-evans.safetensors, TaylorSwiftSDXL.safetensors) name
    UNet layers input_blocks_* / middle_block_* / output_blocks_*; diffusers remaps them on load."""
    src = _adapter(unet, rank=4, alpha=2.0)
    # add one time_emb_proj Linear (kohya trains emb_layers too)
    g = torch.Generator().manual_seed(3)
    tproj = "up_blocks.1.resnets.2.time_emb_proj"
    lin = unet.get_submodule(tproj)
    src.weights[tproj] = (torch.randn(4, lin.in_features, generator=g), torch.randn(lin.out_features, 4, generator=g))
    sd = {}
    for mod, (a, b) in src.weights.items():
        flat = "lora_unet_" + _sgm_flat(mod)
        sd[flat + ".lora_down.weight"], sd[flat + ".lora_up.weight"], sd[flat + ".alpha"] = a, b, torch.tensor(2.0)
    assert any("input_blocks_4_1_transformer_blocks_0_attn1_to_q" in k for k in sd)       # the name format kohya writes
    assert any("middle_block_1_" in k for k in sd) and any("output_blocks_5_1_" in k for k in sd)
    got, skipped = loaders.parse_lora_state_dict(sd, loaders.linear_module_paths(unet), "k")
    assert not skipped and set(got.weights) == set(src.weights)
    for key in list(src.weights)[::5] + [tproj]:
        torch.testing.assert_close(_delta(got, key), _delta(src, key), rtol=1e-6, atol=1e-6)
    # spot-check the published SDXL numbering (layers_per_block = 2)
    f = loaders.sgm_flat_to_diffusers_flat
    assert f("input_blocks_4_1_proj_in") == "down_blocks_1_attentions_0_proj_in"
    assert f("input_blocks_8_1_transformer_blocks_9_ff_net_2") == "down_blocks_2_attentions_1_transformer_blocks_9_ff_net_2"
    assert f("middle_block_1_transformer_blocks_0_attn2_to_v") == "mid_block_attentions_0_transformer_blocks_0_attn2_to_v"
    assert f("output_blocks_0_1_proj_out") == "up_blocks_0_attentions_0_proj_out"
    assert f("output_blocks_5_1_transformer_blocks_1_attn1_to_out_0") == "up_blocks_1_attentions_2_transformer_blocks_1_attn1_to_out_0"
    # a conv target stays an error, now with its diffusers name resolved
    bad = {"lora_unet_input_blocks_1_0_in_layers_2.lora_down.weight": torch.zeros(4, 8), "lora_unet_input_blocks_1_0_in_layers_2.lora_up.weight": torch.zeros(8, 4)}
    with pytest.raises(loaders.LoaderError):
        loaders.parse_lora_state_dict({**sd, **bad}, loaders.linear_module_paths(unet), "k")


@pytest.mark.parametrize("style", ["peft", "diffusers", "kohya"])
def test_text_encoder_halves_are_parsed_not_dropped(unet, style):
    src = _adapter(unet, rank=4)
    g = torch.Generator().manual_seed(1)
    te = {1: {"text_model.encoder.layers.0.self_attn.q_proj": (torch.randn(4, 16, generator=g), torch.randn(16, 4, generator=g)),
              "text_model.encoder.layers.2.mlp.fc1": (torch.randn(4, 16, generator=g), torch.randn(32, 4, generator=g))},
          2: {"text_model.encoder.layers.1.self_attn.out_proj": (torch.randn(4, 24, generator=g), torch.randn(24, 4, generator=g))}}
    src.text_encoder = te
    sd = loaders.lora_state_dict(src, style)
    assert any(k.startswith(("text_encoder_2.", "lora_te2_")) for k in sd)
    got, skipped = loaders.parse_lora_state_dict(sd, loaders.linear_module_paths(unet), "x")
    assert not skipped
    assert {n: set(m) for n, m in got.text_encoder.items()} == {n: set(m) for n, m in te.items()}
    for n in te:
        for mod, (a, b) in te[n].items():
            ga, gb = got.text_encoder[n][mod]
            torch.testing.assert_close(gb @ ga, b @ a, rtol=1e-6, atol=1e-6)


def test_mixed_per_layer_ranks_are_accepted(unet):
    """PEFT rank_pattern / kohya per-layer dims: each layer keeps its own rank; the adapter's width is the largest."""
    g = torch.Generator().manual_seed(0)
    w = {}
    for i, key in enumerate(lora_target_names(unet)):
        r = (4, 8, 16)[i % 3]
        lin = unet.get_submodule(key)
        w[key] = (torch.randn(r, lin.in_features, generator=g), torch.randn(lin.out_features, r, generator=g))
    sd = {}
    for mod, (a, b) in w.items():
        flat = "lora_unet_" + mod.replace(".", "_")
        sd[flat + ".lora_down.weight"], sd[flat + ".lora_up.weight"], sd[flat + ".alpha"] = a, b, torch.tensor(4.0)
    got, _ = loaders.parse_lora_state_dict(sd, loaders.linear_module_paths(unet), "m")
    assert got.rank == 16
    for key, (a, b) in list(w.items())[:6]:
        torch.testing.assert_close(_delta(got, key), (4.0 / a.shape[0]) * (b @ a), rtol=1e-6, atol=1e-6)
    # explicit adapter-level alpha: PEFT's alpha / r uses each layer's own r
    ad = LoraAdapter("m", w, alpha=8.0)
    ks = list(w)
    assert ad.scaling(ks[0]) == 8.0 / 4 and ad.scaling(ks[1]) == 8.0 / 8 and ad.scaling(ks[2]) == 8.0 / 16


def test_ip_adapter_checkpoint_index_follows_diffusers_attn_processor_order():
    """instantid_single_pieline.py:208-212 loads ``ip_adapter`` through ModuleList(unet.attn_processors.values()); diffusers
    registers down_blocks and up_blocks before mid_block, so SDXL's 140 processors are down 0-47, up 48-119, mid 120-139
    (ADVICE r1 / VERDICT weak #4).  mid_block and up_blocks.0 are both 1280 wide: a wrong order would load silently."""
    from omg_amd.ip_adapter import IPAdapter, attn_processor_index
    u = UNet2DConditionModel(UNetConfig.sdxl(), device="meta")
    ix = attn_processor_index(u)
    assert len(ix) == 140
    assert ix["down_blocks.1.attentions.0.transformer_blocks.0.attn1"] == 0
    assert ix["down_blocks.2.attentions.1.transformer_blocks.9.attn2"] == 47
    assert ix["up_blocks.0.attentions.0.transformer_blocks.0.attn1"] == 48
    assert ix["up_blocks.1.attentions.2.transformer_blocks.1.attn2"] == 119
    assert ix["mid_block.attentions.0.transformer_blocks.0.attn1"] == 120
    assert ix["mid_block.attentions.0.transformer_blocks.9.attn2"] == 139
    assert list(u.attn_processors)[48].startswith("up_blocks.0.") and list(u.attn_processors)[120].startswith("mid_block.")
    ipa = IPAdapter(u)
    assert [i for i, _, _ in ipa.layers] == list(range(1, 140, 2))             # every attn2, odd indices
    # a state dict built in diffusers' enumeration order lands on the right modules (tiny UNet on the CPU)
    t = UNet2DConditionModel(UNetConfig.tiny(), dtype=torch.float32, device="cpu")
    ipa = IPAdapter(t)
    names = sorted(attn_processor_index(t).items(), key=lambda kv: kv[1])
    sd, want = {}, {}
    for idx, (name, _) in enumerate(names):
        m = t.get_submodule(name)
        if m.is_cross:
            wk = torch.full((m.inner_dim, m.to_k.in_features), float(idx)); wv = -wk
            sd[f"{idx}.to_k_ip.weight"], sd[f"{idx}.to_v_ip.weight"] = wk, wv
            want[name] = float(idx)
    ipa.load_state_dict({"ip_adapter": sd, "image_proj": {}})
    order = [n.split(".")[0] for n, _ in names]
    assert order == sorted(order, key=["down_blocks", "up_blocks", "mid_block"].index)
    for name, v in want.items():
        m = t.get_submodule(name)
        assert float(m.ip_kv_weight[0, 0]) == v and float(m.ip_kv_weight[m.inner_dim, 0]) == -v
    assert ipa.state_dict().keys() == sd.keys()
    # shapes are validated: swapping two layers of different width is an error, not a silent load
    bad = dict(sd)
    k_small = next(k for k in sd if sd[k].shape[0] != sd["1.to_k_ip.weight"].shape[0])
    bad["1.to_k_ip.weight"] = sd[k_small]
    with pytest.raises(ValueError):
        ipa.load_state_dict({"ip_adapter": bad})
