"""Checkpoint and LoRA file loaders — the on-disk formats on the caller's side of the hot path.

The reference gets its weights through diffusers: ``from_pretrained(..., variant="fp16")`` for the UNet / ControlNet
(/root/reference inference_lora.py:151-157) and ``pipe.load_lora_weights(path, weight_name="pytorch_lora_weights.safetensors",
adapter_name=...)`` per concept and for the optional style LoRA (inference_lora.py:160-169), which accepts three key
styles for the UNet part of a LoRA file:

* PEFT:        ``unet.<module path>.lora_A.weight`` / ``.lora_B.weight``           (diffusers >= 0.26 / peft saves)
* diffusers:   ``unet.<module path>.lora.down.weight`` / ``.lora.up.weight``        (and the attention-processor form
               ``unet.<attn path>.processor.<to_q|to_k|to_v|to_out>_lora.down.weight`` of diffusers <= 0.21)
* kohya-ss:    ``lora_unet_<module path with '.' -> '_'>.lora_down.weight`` / ``.lora_up.weight`` / ``.alpha``, where the
               module path is either diffusers' (``down_blocks_1_attentions_0_...``) or — what kohya-ss's SDXL trainer writes
               and what the concept files the reference ships with use (``chris-evans.safetensors`` etc., inference_lora.py
               ``--lora_path``) — the SGM/ldm block naming ``input_blocks_4_1_...`` / ``middle_block_1_...`` /
               ``output_blocks_3_1_...``, remapped here as diffusers' ``_maybe_map_sgm_blocks_to_diffusers`` does [recalled].

All are mapped onto the module paths of :class:`omg_amd.unet.UNet2DConditionModel` (whose state-dict keys equal
diffusers', boundary B4) and returned as a :class:`omg_amd.lora.LoraAdapter`.  A per-layer ``alpha`` (kohya) is folded
into the up matrix (``B <- B * alpha / r``); per-layer ranks may differ (PEFT ``rank_pattern``).  Text-encoder entries
(``text_encoder.*`` / ``text_encoder_2.*``, ``lora_te1_*`` / ``lora_te2_*``) are parsed into ``adapter.text_encoder[1|2]``
(module paths of transformers' CLIP text models) — the reference encodes every region prompt with the concept LoRA active
on both text encoders (lora_pipeline.py:336-347); :func:`omg_amd.text_encoder.make_encode_prompt` applies them.

Nothing here touches the GPU: tensors stay on the host until ``LoraBank.build`` / ``load_state_dict`` moves them.
"""
from __future__ import annotations

import os
import re
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .lora import LoraAdapter
from .modules import Linear


class LoaderError(ValueError):
    pass


def _read_tensors(path_or_dict) -> Dict[str, torch.Tensor]:
    if isinstance(path_or_dict, dict):
        return path_or_dict
    path = os.fspath(path_or_dict)
    if os.path.isdir(path):          # diffusers passes a directory + weight_name (inference_lora.py:161,167)
        path = os.path.join(path, "pytorch_lora_weights.safetensors")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    obj = torch.load(path, map_location="cpu", weights_only=True)
    return obj.get("state_dict", obj) if isinstance(obj, dict) else obj


# ---------------------------------------------------------------------------------------------- model checkpoints
def load_model_weights(model: torch.nn.Module, path_or_dict, *, strict: bool = True, prefix: str = "",
                       allow_extra: bool = False) -> List[str]:
    """Load a diffusers-layout checkpoint (``diffusion_pytorch_model[.fp16].safetensors`` of a UNet / ControlNet) into
    ``model``.  ``prefix`` strips a leading namespace (``"unet."`` for a whole-pipeline single file).  ``allow_extra``
    accepts a file that holds more than the model (a full VAE file for the decoder-only module) while still requiring every
    tensor of the model.  Shapes are checked before anything is copied; dtype follows the model (fp32 files are rounded
    once).  Returns the ignored keys."""
    sd = _read_tensors(path_or_dict)
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    extra = [k for k in sd if k not in own]
    if strict and (missing or (extra and not allow_extra)):
        raise LoaderError(f"checkpoint does not match {type(model).__name__}: {len(missing)} missing (e.g. {missing[:3]}), "
                          f"{len(extra)} unexpected (e.g. {extra[:3]})")
    bad = [(k, tuple(sd[k].shape), tuple(own[k].shape)) for k in own if k in sd and sd[k].shape != own[k].shape]
    if bad:
        raise LoaderError(f"shape mismatch for {len(bad)} tensors, first: {bad[0]}")
    model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=strict)
    return extra


# ---------------------------------------------------------------------------------------------- LoRA files
_PEFT = re.compile(r"^(?:unet\.)?(?P<mod>.+)\.lora_(?P<ab>[AB])(?:\.[^.]+)?\.weight$")          # optional adapter name segment
_DIFF = re.compile(r"^(?:unet\.)?(?P<mod>.+)\.lora\.(?P<ab>down|up)\.weight$")
_PROC = re.compile(r"^(?:unet\.)?(?P<attn>.+)\.processor\.(?P<proj>to_q|to_k|to_v|to_out)_lora\.(?P<ab>down|up)\.weight$")
_KOHYA = re.compile(r"^lora_unet_(?P<flat>.+)\.(?P<what>lora_down\.weight|lora_up\.weight|alpha)$")
_KOHYA_TE = re.compile(r"^lora_te(?P<n>[12]?)_(?P<flat>.+)\.(?P<what>lora_down\.weight|lora_up\.weight|alpha)$")
_TE_MOD = re.compile(r"^text_encoder(?P<n>_2)?\.(?P<rest>.+)$")
_SGM = re.compile(r"^(?P<blk>input_blocks|output_blocks)_(?P<i>\d+)_(?P<j>\d+)_(?P<rest>.+)$")
_SGM_MID = re.compile(r"^middle_block_(?P<j>\d+)_(?P<rest>.+)$")
_SGM_RES = (("in_layers_0", "norm1"), ("in_layers_2", "conv1"), ("emb_layers_1", "time_emb_proj"), ("out_layers_0", "norm2"),
            ("out_layers_3", "conv2"), ("skip_connection", "conv_shortcut"))


def sgm_flat_to_diffusers_flat(flat: str, layers_per_block: int = 2) -> str:
    """``input_blocks_4_1_transformer_blocks_0_attn1_to_q`` -> ``down_blocks_1_attentions_0_transformer_blocks_0_attn1_to_q``.

    SGM numbering of the SDXL UNet: ``input_blocks.0`` is conv_in; input block ``i >= 1`` is layer ``(i-1) % (lpb+1)`` of down
    block ``(i-1) // (lpb+1)`` (the last layer of a group is the downsampler ``.0.op``); ``middle_block.{0,1,2}`` = resnet,
    transformer, resnet; output block ``i`` is layer ``i % (lpb+1)`` of up block ``i // (lpb+1)``.  Inside a block ``.0`` is the
    resnet and ``.1`` the transformer (``.1`` / ``.2`` of an output block without / with attention is the upsampler ``conv``).
    Names that are not SGM are returned unchanged."""
    n = layers_per_block + 1

    def resnet(rest):
        for a, b in _SGM_RES:
            if rest == a or rest.startswith(a + "_"):
                return b + rest[len(a):]
        return rest

    m = _SGM_MID.match(flat)
    if m:
        j, rest = int(m["j"]), m["rest"]
        if j == 1:
            return f"mid_block_attentions_0_{rest}"
        return f"mid_block_resnets_{j // 2}_{resnet(rest)}"
    m = _SGM.match(flat)
    if not m:
        return flat
    i, j, rest = int(m["i"]), int(m["j"]), m["rest"]
    if m["blk"] == "input_blocks":
        if i == 0:
            return flat
        b, l = (i - 1) // n, (i - 1) % n
        if rest.startswith("op_") or rest == "op":
            return f"down_blocks_{b}_downsamplers_0_conv" + rest[2:]
        return f"down_blocks_{b}_attentions_{l}_{rest}" if j == 1 else f"down_blocks_{b}_resnets_{l}_{resnet(rest)}"
    b, l = i // n, i % n
    if rest.startswith("conv_") or rest == "conv":
        return f"up_blocks_{b}_upsamplers_0_conv" + rest[4:]
    return f"up_blocks_{b}_attentions_{l}_{rest}" if j == 1 else f"up_blocks_{b}_resnets_{l}_{resnet(rest)}"


def linear_module_paths(unet: torch.nn.Module) -> List[str]:
    return [n for n, m in unet.named_modules() if isinstance(m, Linear)]


_TE_PROJ = ("q_proj", "k_proj", "v_proj", "out_proj", "fc1", "fc2")


def _te_unflatten(flat: str) -> str:
    """``text_model_encoder_layers_0_self_attn_q_proj`` -> ``text_model.encoder.layers.0.self_attn.q_proj``."""
    m = re.match(r"^text_model_encoder_layers_(\d+)_(self_attn|mlp)_(\w+)$", flat)
    if not m or m[3] not in _TE_PROJ:
        raise LoaderError(f"unrecognised text-encoder LoRA module {flat!r}")
    return f"text_model.encoder.layers.{m[1]}.{m[2]}.{m[3]}"


def parse_lora_state_dict(sd: Dict[str, torch.Tensor], module_paths: Iterable[str], name: str = "lora",
                          layers_per_block: int = 2) -> Tuple[LoraAdapter, List[str]]:
    """Key-style detection + mapping (see the module docstring).  Returns ``(adapter, skipped_keys)``; raises
    ``LoaderError`` for a UNet entry that names no Linear layer of ``module_paths`` (conv LoRA is not supported by the
    slot GEMM) or for a half-present pair.  ``skipped_keys`` lists what was neither UNet nor text-encoder material."""
    paths = set(module_paths)
    flat = {p.replace(".", "_"): p for p in paths}
    # key = ("unet", module) or ("te", 1 | 2, module)
    down: Dict[tuple, torch.Tensor] = {}
    up: Dict[tuple, torch.Tensor] = {}
    alpha: Dict[tuple, float] = {}
    skipped: List[str] = []

    def put(mod: str, which: str, t: torch.Tensor, key: str):
        if mod.endswith(".to_out") and mod + ".0" in paths:      # diffusers' Attention.to_out is [Linear, Dropout]
            mod = mod + ".0"
        if mod not in paths:
            raise LoaderError(f"LoRA entry {key!r} targets {mod!r}, which is not a Linear layer of this UNet "
                              f"(LoRA on conv layers is not supported)")
        (down if which == "down" else up)[("unet", mod)] = t

    def put_te(n: int, mod: str, which: str, t: torch.Tensor):
        (down if which == "down" else up)[("te", n, mod)] = t

    for key, t in sd.items():
        m = _KOHYA_TE.match(key)
        if m:
            k = ("te", 2 if m["n"] == "2" else 1, _te_unflatten(m["flat"]))
            if m["what"] == "alpha":
                alpha[k] = float(t)
            else:
                put_te(k[1], k[2], "down" if m["what"].startswith("lora_down") else "up", t)
            continue
        m = _TE_MOD.match(key)
        if m:
            n, rest = (2 if m["n"] else 1), m["rest"]
            mm = (re.match(r"^(?P<mod>.+)\.lora_(?P<ab>[AB])(?:\.[^.]+)?\.weight$", rest)
                  or re.match(r"^(?P<mod>.+)\.(?:lora_linear_layer|lora)\.(?P<ab>down|up)\.weight$", rest))
            if mm:
                put_te(n, mm["mod"], {"A": "down", "B": "up"}.get(mm["ab"], mm["ab"]), t)
            else:
                skipped.append(key)
            continue
        m = _PROC.match(key)
        if m:
            put(f"{m['attn']}.{m['proj']}", m["ab"], t, key)
            continue
        m = _DIFF.match(key)
        if m:
            put(m["mod"], m["ab"], t, key)
            continue
        m = _PEFT.match(key)
        if m:
            put(m["mod"], "down" if m["ab"] == "A" else "up", t, key)
            continue
        m = _KOHYA.match(key)
        if m:
            mod = flat.get(m["flat"]) or flat.get(sgm_flat_to_diffusers_flat(m["flat"], layers_per_block))
            if mod is None:
                raise LoaderError(f"kohya LoRA entry {key!r} matches no Linear layer of this UNet "
                                  f"(LoRA on conv layers is not supported)")
            if m["what"] == "alpha":
                alpha[("unet", mod)] = float(t)
            else:
                put(mod, "down" if m["what"].startswith("lora_down") else "up", t, key)
            continue
        skipped.append(key)

    if not any(k[0] == "unet" for k in list(down) + list(up)):
        raise LoaderError("no UNet LoRA entries found (expected PEFT lora_A/lora_B, diffusers lora.down/up or kohya lora_unet_* keys)")
    half = sorted(str(k) for k in set(down) ^ set(up))
    if half:
        raise LoaderError(f"LoRA pair incomplete for {half[:3]} ({len(half)} layers)")
    weights: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
    te: Dict[int, Dict[str, Tuple[torch.Tensor, torch.Tensor]]] = {}
    for k in sorted(down, key=str):
        a, b = down[k].float(), up[k].float()
        if a.dim() != 2 or b.dim() != 2 or b.shape[1] != a.shape[0]:
            raise LoaderError(f"LoRA shapes of {k}: down {tuple(a.shape)}, up {tuple(b.shape)}")
        if k in alpha:
            b = b * (alpha[k] / a.shape[0])
        if k[0] == "unet":
            weights[k[1]] = (a, b)
        else:
            te.setdefault(k[1], {})[k[2]] = (a, b)
    return LoraAdapter(name, weights, alpha=None, text_encoder=te), skipped


def load_lora_adapter(unet: torch.nn.Module, path_or_dict, adapter_name: Optional[str] = None) -> LoraAdapter:
    """``pipe.load_lora_weights(path, adapter_name=...)`` for the UNet half: returns the adapter, ready for
    ``LoraBank(unet, [adapters...]).build(...)``.  The default name follows the reference
    (``lora_path.split('/')[-1].split('.')[0]``, inference_lora.py:166)."""
    if adapter_name is None:
        if isinstance(path_or_dict, dict):
            adapter_name = "lora"
        else:
            adapter_name = os.fspath(path_or_dict).rstrip("/").split("/")[-1].split(".")[0]
    lpb = getattr(getattr(unet, "config", None), "layers_per_block", 2)
    adapter, skipped = parse_lora_state_dict(_read_tensors(path_or_dict), linear_module_paths(unet), adapter_name, lpb)
    adapter.skipped_keys = skipped
    return adapter


# ---------------------------------------------------------------------------------------------- writers (tests, tooling)
def lora_state_dict(adapter: LoraAdapter, style: str = "peft") -> Dict[str, torch.Tensor]:
    """The inverse mapping, in any of the three key styles (used by the round-trip tests and to export synthetic adapters);
    the text-encoder half, when present, is written in the same style."""
    out: Dict[str, torch.Tensor] = {}
    for n, mods in adapter.text_encoder.items():
        pre = "text_encoder" if n == 1 else "text_encoder_2"
        for mod, (a, b) in mods.items():
            a, b = a.contiguous(), b.contiguous()
            if style == "peft":
                out[f"{pre}.{mod}.lora_A.weight"], out[f"{pre}.{mod}.lora_B.weight"] = a, b
            elif style == "diffusers":
                out[f"{pre}.{mod}.lora_linear_layer.down.weight"], out[f"{pre}.{mod}.lora_linear_layer.up.weight"] = a, b
            else:
                flat = f"lora_te{n}_" + mod.replace(".", "_")
                out[f"{flat}.lora_down.weight"], out[f"{flat}.lora_up.weight"] = a, b
                out[f"{flat}.alpha"] = torch.tensor(float(a.shape[0]))
    for mod, (a, b) in adapter.weights.items():
        a, b = a.contiguous(), b.contiguous()
        if style == "peft":
            out[f"unet.{mod}.lora_A.weight"], out[f"unet.{mod}.lora_B.weight"] = a, b
        elif style == "diffusers":
            out[f"unet.{mod}.lora.down.weight"], out[f"unet.{mod}.lora.up.weight"] = a, b
        elif style == "kohya":
            flat = "lora_unet_" + mod.replace(".", "_")
            out[f"{flat}.lora_down.weight"], out[f"{flat}.lora_up.weight"] = a, b
            out[f"{flat}.alpha"] = torch.tensor(float(adapter.alpha if adapter._alpha_given else a.shape[0]))
        else:
            raise ValueError(style)
    return out
