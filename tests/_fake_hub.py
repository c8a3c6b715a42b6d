"""Synthetic model directories in diffusers' on-disk layout (tiny SDXL topology, random weights) for the tests of
omg_amd.compat — no real checkpoint exists offline.  Also LoRA files in the key styles the reference's files use."""
import json
import os

import torch
from safetensors.torch import save_file

WORDS = ["a", "man", "and", "woman", "walking", "on", "the", "street", "dog", "in", "park", "painting", "of"]

UNET_CFG = dict(_class_name="UNet2DConditionModel", in_channels=4, out_channels=4, sample_size=16, block_out_channels=[64, 128, 256],
                down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"], layers_per_block=2,
                transformer_layers_per_block=[1, 1, 2], attention_head_dim=[1, 2, 4], cross_attention_dim=128,
                addition_embed_type="text_time", addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32,
                use_linear_projection=True, norm_num_groups=32, norm_eps=1e-5)


def _rand_state(module, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in module.state_dict().items():
        if k.endswith(".weight") and v.dim() >= 2:
            w = torch.randn(v.shape, generator=g) * v[0].numel() ** -0.5
        elif k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            w = 0.1 * torch.randn(v.shape, generator=g)
        sd[k] = w.to(v.dtype).contiguous()
    return sd


def write_tokenizer(folder):
    os.makedirs(folder, exist_ok=True)
    vocab, merges, i = {}, [], 0
    chars = sorted(set("".join(WORDS)))
    for c in chars + [c + "</w>" for c in chars]:
        vocab[c] = i; i += 1
    for w in WORDS:
        toks = list(w[:-1]) + [w[-1] + "</w>"]
        while len(toks) > 1:
            a, b = toks[0], toks[1]
            if (a, b) not in merges:
                merges.append((a, b))
            if a + b not in vocab:
                vocab[a + b] = i; i += 1
            toks = [a + b] + toks[2:]
    vocab["<|startoftext|>"] = i; i += 1
    vocab["<|endoftext|>"] = i; i += 1
    json.dump(vocab, open(os.path.join(folder, "vocab.json"), "w"))
    open(os.path.join(folder, "merges.txt"), "w").write("#version: 0.2\n" + "\n".join(a + " " + b for a, b in merges) + "\n")
    json.dump({"model_max_length": 77, "pad_token": "<|endoftext|>", "bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>",
               "unk_token": "<|endoftext|>", "tokenizer_class": "CLIPTokenizer"}, open(os.path.join(folder, "tokenizer_config.json"), "w"))
    return len(vocab), vocab["<|endoftext|>"]


def write_sdxl_dir(root, seed=0, plain_copies=False):
    """``plain_copies``: also provide every weight file without the ``.fp16`` variant infix (a real SDXL directory holds both; the
    InstantID script loads its concept pipe without ``variant=``)."""
    from omg_amd.text_encoder import ClipTextConfig, ClipTextEncoder
    from omg_amd.unet import UNet2DConditionModel
    from omg_amd.vae import AutoencoderKLDecoder, VaeConfig
    from omg_amd.compat import unet_config_from_dict
    os.makedirs(root, exist_ok=True)
    json.dump({"_class_name": "StableDiffusionXLPipeline"}, open(os.path.join(root, "model_index.json"), "w"))
    os.makedirs(os.path.join(root, "unet"))
    json.dump(UNET_CFG, open(os.path.join(root, "unet", "config.json"), "w"))
    unet = UNet2DConditionModel(unet_config_from_dict(UNET_CFG), dtype=torch.float16, device="cpu")
    save_file(_rand_state(unet, seed), os.path.join(root, "unet", "diffusion_pytorch_model.fp16.safetensors"))
    os.makedirs(os.path.join(root, "vae"))
    vcfg = dict(_class_name="AutoencoderKL", latent_channels=4, out_channels=3, block_out_channels=[64, 64, 64, 64], layers_per_block=1,
                norm_num_groups=32, scaling_factor=0.13025, force_upcast=True)
    json.dump(vcfg, open(os.path.join(root, "vae", "config.json"), "w"))
    vae = AutoencoderKLDecoder(VaeConfig(block_out_channels=(64, 64, 64, 64), layers_per_block=1), dtype=torch.float16, device="cpu")
    vsd = _rand_state(vae, seed + 1)
    vsd["encoder.conv_in.weight"] = torch.zeros(8, 3, 3, 3, dtype=torch.float16)          # a full VAE file also holds the encoder
    save_file(vsd, os.path.join(root, "vae", "diffusion_pytorch_model.fp16.safetensors"))
    nvocab, eos = write_tokenizer(os.path.join(root, "tokenizer"))
    write_tokenizer(os.path.join(root, "tokenizer_2"))
    for sub, act, proj, s in (("text_encoder", "quick_gelu", False, 2), ("text_encoder_2", "gelu", True, 3)):
        os.makedirs(os.path.join(root, sub))
        c = dict(architectures=["CLIPTextModelWithProjection" if proj else "CLIPTextModel"], vocab_size=nvocab, hidden_size=64, intermediate_size=128,
                 num_hidden_layers=2, num_attention_heads=1, max_position_embeddings=77, hidden_act=act, layer_norm_eps=1e-5, projection_dim=64,
                 eos_token_id=eos)
        json.dump(c, open(os.path.join(root, sub, "config.json"), "w"))
        enc = ClipTextEncoder(ClipTextConfig(vocab_size=nvocab, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                                             hidden_act=act, projection_dim=64, eos_token_id=eos, with_projection=proj), dtype=torch.float16, device="cpu")
        sd = _rand_state(enc, seed + s)
        sd["text_model.embeddings.position_ids"] = torch.arange(77)[None]                 # older transformers files carry this buffer
        save_file(sd, os.path.join(root, sub, "model.fp16.safetensors"))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump({"_class_name": "EulerDiscreteScheduler"}, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    if plain_copies:
        for sub, stem in (("unet", "diffusion_pytorch_model"), ("vae", "diffusion_pytorch_model"), ("text_encoder", "model"), ("text_encoder_2", "model")):
            os.link(os.path.join(root, sub, stem + ".fp16.safetensors"), os.path.join(root, sub, stem + ".safetensors"))
    return root


def write_controlnet_dir(root, seed=7):
    from omg_amd.compat import unet_config_from_dict
    from omg_amd.controlnet import ControlNetModel
    os.makedirs(root, exist_ok=True)
    c = dict(UNET_CFG); c["_class_name"] = "ControlNetModel"; c["conditioning_channels"] = 3; c.pop("up_block_types")
    json.dump(c, open(os.path.join(root, "config.json"), "w"))
    net = ControlNetModel(unet_config_from_dict(c), dtype=torch.float16, device="cpu")
    sd = _rand_state(net, seed)
    for k in sd:                                                                          # zero-convs of a trained ControlNet are not zero: keep random
        pass
    save_file(sd, os.path.join(root, "diffusion_pytorch_model.safetensors"))
    return root


def sgm_flat(mod: str) -> str:
    import re
    m = re.match(r"^down_blocks\.(\d+)\.(attentions|resnets)\.(\d+)\.(.*)$", mod)
    if m:
        head = f"input_blocks.{1 + 3 * int(m[1]) + int(m[3])}.{1 if m[2] == 'attentions' else 0}."
        rest = m[4]
    else:
        m = re.match(r"^up_blocks\.(\d+)\.(attentions|resnets)\.(\d+)\.(.*)$", mod)
        if m:
            head = f"output_blocks.{3 * int(m[1]) + int(m[3])}.{1 if m[2] == 'attentions' else 0}."
            rest = m[4]
        else:
            m = re.match(r"^mid_block\.(attentions|resnets)\.(\d+)\.(.*)$", mod)
            head = f"middle_block.{1 if m[1] == 'attentions' else 2 * int(m[2])}."
            rest = m[3]
    return (head + rest).replace(".", "_")


def write_lora_file(path, unet, seed, rank=4, style="kohya", text_encoders=None):
    """A concept LoRA file: kohya-ss SDXL keys (SGM block names, per-layer alpha, lora_te1/lora_te2) or PEFT keys."""
    from omg_amd.lora import lora_target_names
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for mod in lora_target_names(unet):
        lin = unet.get_submodule(mod)
        a = torch.randn(rank, lin.in_features, generator=g) * lin.in_features ** -0.5
        b = torch.randn(lin.out_features, rank, generator=g) * 0.1
        if style == "kohya":
            f = "lora_unet_" + sgm_flat(mod)
            sd[f + ".lora_down.weight"], sd[f + ".lora_up.weight"], sd[f + ".alpha"] = a, b, torch.tensor(float(rank))
        else:
            sd[f"unet.{mod}.lora_A.weight"], sd[f"unet.{mod}.lora_B.weight"] = a, b
    for n, enc in enumerate(text_encoders or [], start=1):
        for li in range(len(enc.text_model.encoder.layers)):
            for sub in ("self_attn.q_proj", "self_attn.v_proj", "mlp.fc1"):
                lin = enc.text_model.encoder.layers[li].get_submodule(sub)
                a = torch.randn(rank, lin.in_features, generator=g) * lin.in_features ** -0.5
                b = torch.randn(lin.out_features, rank, generator=g) * 0.3
                if style == "kohya":
                    f = f"lora_te{n}_text_model_encoder_layers_{li}_" + sub.replace(".", "_")
                    sd[f + ".lora_down.weight"], sd[f + ".lora_up.weight"], sd[f + ".alpha"] = a, b, torch.tensor(float(rank))
                else:
                    pre = "text_encoder" if n == 1 else "text_encoder_2"
                    sd[f"{pre}.text_model.encoder.layers.{li}.{sub}.lora_A.weight"], sd[f"{pre}.text_model.encoder.layers.{li}.{sub}.lora_B.weight"] = a, b
    os.makedirs(os.path.dirname(path), exist_ok=True)
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    return path
