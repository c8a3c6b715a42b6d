"""ORACLE (test infrastructure): literal CPU restatement of one OMG denoising iteration and of
the whole stage-1 / stage-2 loop.

Follows /root/reference ``src/pipelines/lora_pipeline.py``:
  :397-409  latents duplicated x2                :467-474  CFG concat order [neg, pos]
  :491-492  cat([latents]*2) + scale_model_input  :546-566  main UNet call
  :568-607  region fusion (i > 15 and stage == 2) :674-681  get_region_mask (union, nearest resize)
  :583-599  concept pass on latent_model_input[3:4] duplicated, per-concept prompt embeddings
  :610-612  classifier-free guidance              :615      scheduler.step
(identical block: src/pipelines/instantid_pipeline.py:618-707).

PINNED (round 5) BY THE REFERENCE'S OWN LOOP RUN IN THE BUILD CONTAINER: tests/golden/make_golden_loop.py imports
/root/reference/src/pipelines/lora_pipeline.py and instantid_pipeline.py unmodified under stand-in diffusers / torchvision modules and
records the per-step latents of LoraMultiConceptPipeline.__call__ / InstantidMultiConceptPipeline.__call__ (ten cases: overlapping
masks, a None mask, all masks None, one and three concepts, styleL, non-square latents, a ControlNet, InstantID with / without the t2i ControlNet) in
tests/golden/loop_golden.npz; tests/test_oracle_loop.py reproduces them with `denoise` below — stage 1 bit-equal, stage 2 within one
fp32 ulp.  (The reference ships no tests or recorded outputs of its own.)  float32 torch on CPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.nn.functional as F

FUSION_START = 15  # `if i > 15 and stage == 2` (lora_pipeline.py:568)


def region_union(masks: Sequence[Optional[torch.Tensor]], h: int, w: int) -> torch.Tensor:
    """get_region_mask (lora_pipeline.py:674-681)."""
    excl = torch.zeros((h, w))
    for m in masks:
        if m is not None:
            r = F.interpolate(m[None, None].float(), size=(h, w), mode="nearest").squeeze()
            excl = ((r == 1) | (excl == 1)).to(r.dtype)
    return excl


def fuse_noise(noise_pred: torch.Tensor, region_preds: Sequence[Optional[torch.Tensor]],
               masks: Sequence[Optional[torch.Tensor]]) -> torch.Tensor:
    """lora_pipeline.py:569-607.  noise_pred (4,C,H,W) = [unc0, unc1, cond0, cond1]; each region
    prediction (2,C,H,W) = [unc, cond].  Returns the updated noise_pred (copy)."""
    noise_pred = noise_pred.clone()
    h, w = noise_pred.shape[2:]
    union = region_union(masks, h, w)
    edit = torch.cat([noise_pred[1:2], noise_pred[3:4]], dim=0)
    new = torch.zeros_like(edit)
    new[:, :, union == 0] = edit[:, :, union == 0]
    replace_ratio = 1.0
    new[:, :, union != 0] = (1 - replace_ratio) * edit[:, :, union != 0]
    for rp, m in zip(region_preds, masks):
        if m is None:
            continue
        cm = F.interpolate(m[None, None].float(), size=(h, w), mode="nearest").squeeze()
        sel = cm == 1
        new[:, :, sel] += replace_ratio * (rp[:, :, sel] / cm.reshape(1, 1, h, w)[:, :, sel])
    noise_pred[1] = new[0]
    noise_pred[3] = new[1]
    return noise_pred


def cfg(noise_pred: torch.Tensor, guidance_scale: float) -> torch.Tensor:
    unc, txt = noise_pred.chunk(2)
    return unc + guidance_scale * (txt - unc)


def denoise(main_unet: Callable, concept_unets: Sequence[Optional[Callable]], scheduler, latents0: torch.Tensor,
            n_steps: int, guidance_scale: float, stage: int, masks: Sequence[Optional[torch.Tensor]] = (),
            fusion_start: int = FUSION_START, record: Optional[list] = None) -> torch.Tensor:
    """The loop at lora_pipeline.py:485-632.

    main_unet(x[4,C,H,W], i) -> noise[4,C,H,W]; concept_unets[c](x[2,C,H,W], i) -> noise[2,C,H,W].
    latents0: (1,C,H,W) already multiplied by init_noise_sigma.  Returns final latents (2,C,H,W)."""
    latents = torch.cat([latents0, latents0.clone()]).float()
    for i in range(n_steps):
        x = scheduler.scale_model_input(torch.cat([latents] * 2), i).float()
        noise = main_unet(x, i)
        if i > fusion_start and stage == 2:
            regs: List[Optional[torch.Tensor]] = []
            for unet_c, m in zip(concept_unets, masks):
                regs.append(unet_c(torch.cat([x[3:4]] * 2), i) if m is not None else None)
            noise = fuse_noise(noise, regs, masks)
        eps = cfg(noise, guidance_scale)
        latents = scheduler.step(eps.double().numpy(), i, latents.double().numpy())
        latents = torch.from_numpy(latents).float()
        if record is not None:
            record.append(latents.clone())
    return latents
