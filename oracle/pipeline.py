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


# ---------------------------------------------------------------------------------------------------------------------------------
# ControlNet call pattern on the MAIN rows (lora_pipeline.py:275-286, :421-428, :497-536; twin instantid_pipeline.py:477-483, :552-589)


def controlnet_keep(n_steps: int, control_guidance_start, control_guidance_end, n_nets: int = 1):
    """lora_pipeline.py:275-286 (scalars broadcast to one entry per ControlNet) + :421-428: ``keep[i][k] = 1 - float(i / S < start_k or (i + 1) / S > end_k)``.
    Returns a list over steps of lists over nets (the reference collapses the inner list to a scalar for a single ControlNetModel)."""
    s, e = control_guidance_start, control_guidance_end
    if not isinstance(s, list) and isinstance(e, list):
        s = len(e) * [s]
    elif not isinstance(e, list) and isinstance(s, list):
        e = len(s) * [e]
    elif not isinstance(s, list) and not isinstance(e, list):
        s, e = n_nets * [s], n_nets * [e]
    return [[1.0 - float(i / n_steps < s_ or (i + 1) / n_steps > e_) for s_, e_ in zip(s, e)] for i in range(n_steps)]


def main_controlnet_residuals(nets: Sequence[Callable], x4: torch.Tensor, i: int, ctx4: torch.Tensor, te4: torch.Tensor, tid4: torch.Tensor,
                              images: Sequence[torch.Tensor], scales: Sequence[float], keep_i: Sequence[float], guess_mode: bool = False):
    """The ControlNet block of one iteration (lora_pipeline.py:497-536) for ONE net or a MultiControlNetModel (diffusers: the nets' residuals are summed
    in list order).  ``nets[k](x, i, ctx, image, scale, te, tid, guess_mode) -> (down[9], mid)``; ``x4`` = the scaled ``[unc0, unc1, cond0, cond1]`` input.
    guess_mode (:497-503, :531-535): the nets see only the CONDITIONAL half — ``scale_model_input(latents)`` is rows 2, 3 of x4, the positive prompt
    embeddings, ``add_text_embeds.chunk(2)[1]``, an image that ``prepare_image`` did not duplicate — and zeros are concatenated in front for the
    unconditional rows.  ``images[k]`` is (1, 3, H, W)."""
    down = mid = None
    for k, net in enumerate(nets):
        sc = scales[k] * keep_i[k]                                            # :511-517
        if guess_mode:
            d, m = net(x4[2:4], i, ctx4[2:4], images[k].repeat(2, 1, 1, 1), sc, te4[2:4], tid4[2:4], True)
        else:
            d, m = net(x4, i, ctx4, images[k].repeat(4, 1, 1, 1), sc, te4, tid4, False)
        if down is None:
            down, mid = list(d), m
        else:                                                                 # MultiControlNetModel.forward: samples_prev + samples_curr
            down = [a + b for a, b in zip(down, d)]
            mid = mid + m
    if guess_mode:                                                            # :531-535
        down = [torch.cat([torch.zeros_like(d), d]) for d in down]
        mid = torch.cat([torch.zeros_like(mid), mid])
    return down, mid
