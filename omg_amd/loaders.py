"""Checkpoint and LoRA file loaders — the on-disk formats on the caller's side of the hot path.

The reference gets its weights through diffusers: ``from_pretrained(..., variant="fp16")`` for the UNet / ControlNet
(/root/reference inference_lora.py:151-157) and ``pipe.load_lora_weights(path, weight_name="pytorch_lora_weights.safetensors",
adapter_name=...)`` per concept and for the optional style LoRA (inference_lora.py:160-169), which accepts three key
styles for the UNet part of a LoRA file:

* PEFT:        ``unet.<module path>.lora_A.weight`` / ``.lora_B.weight``           (diffusers >= 0.26 / peft saves)
* diffusers:   ``unet.<module path>.lora.down.weight`` / ``.lora.up.weight``        (and the attention-processor form
               ``unet.<attn path>.processor.<to_q|to_k|to_v|to_out>_lora.down.weight`` of diffusers <= 0.21)
* kohya-ss:    ``lora_unet_<module path with '.' -> '_'>.lora_down.weight`` / ``.lora_up.weight`` / ``.alpha``

All three are mapped onto the module paths of :class:`omg_amd.unet.UNet2DConditionModel` (whose state-dict keys equal
diffusers', boundary B4) and returned as a :class:`omg_amd.lora.LoraAdapter`.  A per-layer ``alpha`` (kohya) is folded
into the up matrix (``B <- B * alpha / r``) so that the adapter's own ``alpha`` is its rank, which is what
``LoraBank.build`` expects.  Text-encoder entries (``text_encoder.*``, ``lora_te*``) are not part of this path — the
pipeline takes prompt embeddings — and are reported, not loaded.

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


def linear_module_paths(unet: torch.nn.Module) -> List[str]:
    return [n for n, m in unet.named_modules() if isinstance(m, Linear)]


def parse_lora_state_dict(sd: Dict[str, torch.Tensor], module_paths: Iterable[str], name: str = "lora"
                          ) -> Tuple[LoraAdapter, List[str]]:
    """Key-style detection + mapping (see the module docstring).  Returns ``(adapter, skipped_keys)``; raises
    ``LoaderError`` for a UNet entry that names no Linear layer of ``module_paths`` (conv LoRA is not supported by the
    slot GEMM), for a half-present pair, or for mixed ranks."""
    paths = set(module_paths)
    flat = {p.replace(".", "_"): p for p in paths}
    down: Dict[str, torch.Tensor] = {}
    up: Dict[str, torch.Tensor] = {}
    alpha: Dict[str, float] = {}
    skipped: List[str] = []

    def put(mod: str, which: str, t: torch.Tensor, key: str):
        if mod.endswith(".to_out") and mod + ".0" in paths:      # diffusers' Attention.to_out is [Linear, Dropout]
            mod = mod + ".0"
        if mod not in paths:
            raise LoaderError(f"LoRA entry {key!r} targets {mod!r}, which is not a Linear layer of this UNet "
                              f"(LoRA on conv layers is not supported)")
        (down if which == "down" else up)[mod] = t

    for key, t in sd.items():
        if key.startswith(("text_encoder", "lora_te")):
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
            mod = flat.get(m["flat"])
            if mod is None:
                raise LoaderError(f"kohya LoRA entry {key!r} matches no Linear layer of this UNet")
            if m["what"] == "alpha":
                alpha[mod] = float(t)
            else:
                put(mod, "down" if m["what"].startswith("lora_down") else "up", t, key)
            continue
        skipped.append(key)

    if not down and not up:
        raise LoaderError("no UNet LoRA entries found (expected PEFT lora_A/lora_B, diffusers lora.down/up or kohya lora_unet_* keys)")
    half = sorted(set(down) ^ set(up))
    if half:
        raise LoaderError(f"LoRA pair incomplete for {half[:3]} ({len(half)} layers)")
    weights: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
    ranks = set()
    for mod in sorted(down):
        a, b = down[mod].float(), up[mod].float()
        if a.dim() != 2 or b.dim() != 2 or b.shape[1] != a.shape[0]:
            raise LoaderError(f"LoRA shapes of {mod}: down {tuple(a.shape)}, up {tuple(b.shape)}")
        r = a.shape[0]
        ranks.add(r)
        if mod in alpha:
            b = b * (alpha[mod] / r)
        weights[mod] = (a, b)
    if len(ranks) != 1:
        raise LoaderError(f"per-layer ranks differ ({sorted(ranks)}); one rank per adapter is supported")
    return LoraAdapter(name, weights, alpha=None), skipped


def load_lora_adapter(unet: torch.nn.Module, path_or_dict, adapter_name: Optional[str] = None) -> LoraAdapter:
    """``pipe.load_lora_weights(path, adapter_name=...)`` for the UNet half: returns the adapter, ready for
    ``LoraBank(unet, [adapters...]).build(...)``.  The default name follows the reference
    (``lora_path.split('/')[-1].split('.')[0]``, inference_lora.py:166)."""
    if adapter_name is None:
        if isinstance(path_or_dict, dict):
            adapter_name = "lora"
        else:
            adapter_name = os.fspath(path_or_dict).rstrip("/").split("/")[-1].split(".")[0]
    adapter, _ = parse_lora_state_dict(_read_tensors(path_or_dict), linear_module_paths(unet), adapter_name)
    return adapter


# ---------------------------------------------------------------------------------------------- writers (tests, tooling)
def lora_state_dict(adapter: LoraAdapter, style: str = "peft") -> Dict[str, torch.Tensor]:
    """The inverse mapping, in any of the three key styles (used by the round-trip tests and to export synthetic adapters)."""
    out: Dict[str, torch.Tensor] = {}
    for mod, (a, b) in adapter.weights.items():
        a, b = a.contiguous(), b.contiguous()
        if style == "peft":
            out[f"unet.{mod}.lora_A.weight"], out[f"unet.{mod}.lora_B.weight"] = a, b
        elif style == "diffusers":
            out[f"unet.{mod}.lora.down.weight"], out[f"unet.{mod}.lora.up.weight"] = a, b
        elif style == "kohya":
            flat = "lora_unet_" + mod.replace(".", "_")
            out[f"{flat}.lora_down.weight"], out[f"{flat}.lora_up.weight"] = a, b
            out[f"{flat}.alpha"] = torch.tensor(float(adapter.alpha))
        else:
            raise ValueError(style)
    return out
