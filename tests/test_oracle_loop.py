"""CPU tests (-m "not gpu"): the oracle's LOOP (oracle/pipeline.py denoise / fuse_noise / region_union / cfg) and its attention-processor
sequence (oracle/controller.py reference_attn_fn + AttentionReplaceOracle) against trajectories produced by EXECUTING THE REFERENCE'S OWN
``LoraMultiConceptPipeline.__call__`` / ``RegionControlNet_AttnProcessor`` / ``revise_regionally_controlnet_forward`` /
``get_region_mask`` / ``AttentionReplace`` in the build container (tests/golden/make_golden_loop.py -> loop_golden.npz).

This is the pin of SURVEY §8 rows A1, A3, A7, A8, A9 (and of A11's adapter bookkeeping: set_adapters, [lora, "style"] at [0.7, 0.5]):
the GPU loop tests compare the HIP pipeline with exactly these oracle functions.  The ``lora_cn`` / ``iid`` / ``iid_t2i`` cases pin the call
pattern of A13 (which rows a ControlNet sees, with which context and scale: lora_pipeline.py:519-566, instantid_pipeline.py:574-592, :638-657) and
of the InstantID twin of the loop (instantid_pipeline.py:540-707) with the reference's own InstantidSingleConceptPipeline.load_ip_adapter_instantid /
_encode_prompt_image_emb, Resampler and IPAttnProcessor2_0 on the concept side — the ControlNet's and the UNet's own arithmetic stay the oracle's."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import controller as oc
from oracle import controlnet as ocn
from oracle import ip_adapter as oip
from oracle import pipeline as opipe
from oracle import resampler as orsm
from oracle import schedulers as osched
from oracle import unet as ou

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
import make_golden_loop as mk  # noqa: E402   (only `build`, CASES and the prompt strings: no reference import happens at module import)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "loop_golden.npz"))


def oracle_run(c, stage, num_att_layers):
    cfg, sd, table, S = c["cfg"], c["sd"], c["table"], c["steps"]
    osch = osched.make(c["sched"], S)
    octl = oc.AttentionReplaceOracle(*c["ctl_args"])
    octl.num_att_layers = num_att_layers
    attn = oc.reference_attn_fn(octl)
    # lora_pipeline.py:467-474: [neg, neg, pos, pos]; time ids repeated per row (:474)
    ctx4 = torch.stack([table[mk.NEG][0]] * 2 + [table[mk.P][0]] * 2)
    te4 = torch.stack([table[mk.NEG][1]] * 2 + [table[mk.P][1]] * 2)
    tid = torch.tensor([[c["H"], c["W"], 0, 0, c["H"], c["W"]]], dtype=torch.float32)
    names = ou.lora_target_names(cfg)
    fn = {k: ou.make_lora(cfg, names, rank=mk.LORA_RANK, seed=mk.LORA_SEED0 + (50 if k == "style" else int(k[1:])), scale=mk.LORA_SCALE)[1] for k in c["loras"]}
    main_lora = fn["style"] if c["style"] else None              # inference_lora.py:162-164 + the main call's scale 0.8 (:546-566)

    flow = c["flow"]
    cn_kw = mk.CN_VARIANTS.get(flow, {})
    if flow.startswith("lora_cn"):
        # lora_pipeline.py:519-536: the ControlNet sees the four main rows with the main context; its residuals feed the main UNet only.  Round 6: + the guidance
        # window (controlnet_keep, :421-428, :511-517), guess_mode (:497-503, :531-535) and a list of nets (MultiControlNetModel) — oracle/pipeline.py
        multi = flow == "lora_cn_multi"
        csds = [c["csd"], c["csd2"]] if multi else [c["csd"]]
        images = [c["pose"], c["pose2"]] if multi else [c["pose"]]
        scales = [mk.CN_SCALE, mk.T2I_SCALE] if multi else [mk.CN_SCALE]
        keep = opipe.controlnet_keep(S, cn_kw.get("control_guidance_start", 0.0), cn_kw.get("control_guidance_end", 1.0), len(csds))
        nets = [lambda x, i, ctx, img, sc, te, tid_, gm, csd=csd: ocn.controlnet_forward(csd, cfg, x, float(osch.timesteps[i]), ctx, img, sc, te, tid_, guess_mode=gm)
                for csd in csds]

    def main(x, i):
        t = float(osch.timesteps[i])
        down = mid = None
        if flow.startswith("lora_cn"):
            down, mid = opipe.main_controlnet_residuals(nets, x, i, ctx4, te4, tid.repeat(4, 1), images, scales, keep[i], cn_kw.get("guess_mode", False))
        if flow.startswith("iid_t2i"):        # instantid_pipeline.py:574-592: self.controlnet2 on the main rows with t2i_image, scale x controlnet_keep[i] (:578)
            down, mid = ocn.controlnet_forward(c["csd2"], cfg, x, t, ctx4, c["pose2"].repeat(4, 1, 1, 1), mk.T2I_SCALE * iid_keep[i][0], te4, tid.repeat(4, 1))
        return ou.unet_forward(sd, cfg, x, t, ctx4, te4, tid.repeat(4, 1), attn_fn=attn, lora=main_lora,
                               down_block_additional_residuals=down, mid_block_additional_residual=mid)

    iid_keep = opipe.controlnet_keep(S, cn_kw.get("control_guidance_start", 0.0), cn_kw.get("control_guidance_end", 1.0), 1)
    if flow.startswith("iid"):
        ip_fn = oip.make_ip_attn_fn(c["ipw"], mk.IP_SCALE, mk.IP_TOKENS)
        # instantid_single_pieline.py:221-243: [zeros | face embedding] through the Resampler -> (2, 16, D) image-prompt tokens
        faces = []
        for e in c["face_emb"]:
            e = torch.from_numpy(e).reshape(1, 1, mk.FACE_DIM)
            faces.append(orsm.resampler_forward(c["rsd"], torch.cat([torch.zeros_like(e), e]), mk.RESAMPLER["heads"]))

    def conc(k):
        rp, rn = mk.REGION[k]
        ctx2 = torch.stack([table[rn][0], table[rp][0]])          # :345: [negative, positive]
        te2 = torch.stack([table[rn][1], table[rp][1]])
        if flow.startswith("iid"):
            def f(x, i):             # instantid_pipeline.py:638-674: IdentityNet(latents, face tokens, key points) -> concept UNet(text + face tokens), no LoRA scale
                t = float(osch.timesteps[i])
                down, mid = ocn.controlnet_forward(c["csd"], cfg, x, t, faces[k], c["pose"].repeat(2, 1, 1, 1), mk.IDN_SCALE * iid_keep[i][0], te2, tid.repeat(2, 1))      # :566-572
                return ou.unet_forward(sd, cfg, x, t, torch.cat([ctx2, faces[k]], dim=1), te2, tid.repeat(2, 1), attn_fn=ip_fn,
                                       down_block_additional_residuals=down, mid_block_additional_residual=mid)
            return f
        if c["style"]:                                            # :588-589: set_adapters([lora, "style"], adapter_weights=[0.7, 0.5])
            lora = lambda key, x: 0.7 * fn[f"c{k}"](key, x) + 0.5 * fn["style"](key, x)
        else:
            lora = fn[f"c{k}"]
        return lambda x, i: ou.unet_forward(sd, cfg, x, float(osch.timesteps[i]), ctx2, te2, tid.repeat(2, 1), lora=lora)

    rec = []
    opipe.denoise(main, [conc(k) for k in range(c["K"])], osch, c["lat0"] * osch.init_noise_sigma, S, c["gs"], stage, masks=c["masks"], record=rec)
    assert (octl.cur_step, octl.cur_att_layer) == (S, 0)
    return torch.stack(rec).numpy()


@pytest.mark.parametrize("case", mk.CASES, ids=[c[0] for c in mk.CASES])
def test_oracle_loop_matches_the_reference_pipeline_run_here(gold, case):
    c = mk.build(case)
    nl = int(gold[f"{c['name']}/num_att_layers"])
    assert nl == 2 * ou.count_attention_layers(c["cfg"])          # lora_pipeline.py:152
    for stage in (1, 2):
        ref = gold[f"{c['name']}/stage{stage}"]
        got = oracle_run(c, stage, nl)
        assert got.shape == ref.shape == (c["steps"], 2, 4) + tuple(c["lat0"].shape[2:])
        err = np.abs(got - ref).reshape(c["steps"], -1).max(axis=1)
        rms = float(np.sqrt((ref[-1] ** 2).mean()))
        print(f"{c['name']} stage {stage}: max|d| per step first/last = {err[0]:.2e} / {err[-1]:.2e}, latent rms {rms:.3f}")
        # fp32 on both sides; the only differences are summation order (SDPA vs softmax-bmm in the concept pass, einsum vs expand in the
        # controller) amplified by CFG over <= 20 steps
        assert err.max() < 2e-6 * max(1.0, rms), err            # measured: stage 1 bit-equal, stage 2 <= 7.6e-6 at latent rms 16
    s1, s2 = gold[f"{c['name']}/stage1"], gold[f"{c['name']}/stage2"]
    assert np.array_equal(s1[:16], s2[:16]), "stage 2 equals stage 1 until the first fused step (i > 15, lora_pipeline.py:568)"
    assert np.array_equal(s2[:, 0], s1[:, 0]), "the base sample never depends on the edit"
    if any(m is not None for m in c["masks"]):
        assert np.abs(s2[-1][1] - s1[-1][1]).max() > 0.1
    else:
        assert np.array_equal(s2, s1), "all masks None: the union is empty, no concept pass runs, stage 2 is stage 1"
    # one set_adapters per concept at prompt encoding (:340-342) + one per concept WITH a mask per fused step (:588-591), both stages
    n_masked = sum(m is not None for m in c["masks"])
    fused = c["steps"] - 16
    if c["flow"].startswith("iid"):      # the InstantID loop never selects adapters; IdentityNet once per masked concept per fused step (stage 2 only)
        assert int(gold[f"{c['name']}/set_adapters_calls"]) == 0
        assert int(gold[f"{c['name']}/controlnet_calls"]) == n_masked * fused
        assert int(gold[f"{c['name']}/controlnet2_calls"]) == (2 * c["steps"] if c["flow"].startswith("iid_t2i") else 0)
        if f"{c['name']}/controlnet2_scales_seen" in gold.files:      # the one window on both nets (instantid_pipeline.py:566-578)
            kw = mk.CN_VARIANTS[c["flow"]]
            keep = [k_[0] for k_ in opipe.controlnet_keep(c["steps"], kw["control_guidance_start"], kw["control_guidance_end"], 1)]
            assert np.array_equal(gold[f"{c['name']}/controlnet2_scales_seen"], np.array([mk.T2I_SCALE * k_ for k_ in keep] * 2))
            assert np.array_equal(gold[f"{c['name']}/controlnet_scales_seen"], np.array([mk.IDN_SCALE * keep[i] for i in range(16, c["steps"]) for _ in range(n_masked)]))
            assert 0.0 in keep[16:] and 1.0 in keep[16:]
    else:
        assert int(gold[f"{c['name']}/set_adapters_calls"]) == 2 * c["K"] + n_masked * fused
        assert int(gold[f"{c['name']}/controlnet_calls"]) == (2 * c["steps"] if c["flow"].startswith("lora_cn") else 0)      # called every step, also at keep = 0
        if f"{c['name']}/controlnet_scales_seen" in gold.files:      # what the reference handed every net: scale_k * controlnet_keep[i][k], stage 1 then stage 2
            seen = gold[f"{c['name']}/controlnet_scales_seen"]
            kw = mk.CN_VARIANTS[c["flow"]]
            keep = opipe.controlnet_keep(c["steps"], kw.get("control_guidance_start", 0.0), kw.get("control_guidance_end", 1.0), seen.shape[0])
            sc = [mk.CN_SCALE, mk.T2I_SCALE][: seen.shape[0]]
            want = np.array([[sc[k] * keep[i][k] for i in range(c["steps"])] * 2 for k in range(seen.shape[0])])
            assert np.array_equal(seen, want) and (want == 0).any() and (want != 0).any()
