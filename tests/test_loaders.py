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
    sd["text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_A.weight"] = torch.zeros(8, 4)   # ignored, reported
    from safetensors.torch import save_file
    d = tmp_path / "my_concept"
    os.makedirs(d)
    save_file(sd, str(d / "pytorch_lora_weights.safetensors"))
    got = loaders.load_lora_adapter(unet, str(d))                   # directory + default weight name, as the reference calls it
    assert got.name == "my_concept" and got.rank == 8
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
    with pytest.raises(loaders.LoaderError, match="ranks differ"):
        loaders.parse_lora_state_dict({f"unet.{k}.lora_A.weight": a, f"unet.{k}.lora_B.weight": b,
                                       f"unet.{k2}.lora_A.weight": a2, f"unet.{k2}.lora_B.weight": b2}, paths)


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
