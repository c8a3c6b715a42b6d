"""BASELINE configs[0] END TO END AT FULL WIDTH (VERDICT r5 "missing 1" / "next 1"; SURVEY §8(d): "time C1 (512^2, 10 steps) end-to-end
for real").

"SDXL-base 512x512, 10 DDIM steps, 1 concept, no LoRA, CPU diffusers reference (plumbing only)": the reference call is
``LoraMultiConceptPipeline.__call__`` (/root/reference src/pipelines/lora_pipeline.py:485-632) with ``num_inference_steps=10`` — the fusion
branch ``i > 15 and stage == 2`` (:568) never fires, so the stage-2 call must equal the stage-1 call.  It is the ONE BASELINE configuration
whose whole loop the fp32 oracle can execute at the full 2.57 B-parameter width in minutes (10 steps x B = 4 rows at latent 64^2 = 40
sample-forwards of 1.59 TFLOP).  Three trajectories on the same fp16-rounded weights and inputs:

    HIP pipeline (fp16 storage, hipGraph-less eager loop, p2p controller installed)            -- the product
    oracle/pipeline.denoise + oracle/unet.py + oracle/controller.py in fp32 on the host          -- exact arithmetic; its WALL TIME is the
                                                                                                   "reference CPU path" of configs[0], timed for real
    the same under oracle/precision.rounding(torch.float16)                                      -- the reference's own fp16 eager arithmetic

and the per-step distances between them (max / rms over the latent rms).  slow-marked (~12 min of host cores): OMG_RUN_SLOW=1; the result
is tracked as profiles/r06_config0_fullwidth_loop.json and read by bench.py (`cpu_baseline_config0`)."""
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


@pytest.mark.slow
def test_config0_ten_step_loop_at_full_width_matches_the_oracle(dev):
    from omg_amd import controller as pc
    from omg_amd import synthetic
    from omg_amd.pipeline import ConceptModels, LoraMultiConceptPipeline, revise_regionally_controlnet_forward
    from omg_amd.schedulers import make_scheduler
    from omg_amd.unet import UNet2DConditionModel, UNetConfig
    from oracle import controller as oc
    from oracle import pipeline as opipe
    from oracle import precision as oprec
    from oracle import schedulers as osched
    from oracle import unet as ou

    dtype, HW, S, gs = torch.float16, 512, 10, 7.5
    cfg, ocfg = UNetConfig.sdxl(), ou.UNetConfig.sdxl()
    unet = UNet2DConditionModel(cfg, dtype=dtype, device=dev).init_synthetic_(seed=0)
    req = synthetic.c2_inputs(unet, seed=0, n_concepts=1, height=HW, width=HW)
    mask = synthetic.c2_masks(HW, HW)[0]
    req["region_masks"] = [mask]
    P = "a man and a woman walking on the street"
    args = ([P, P], S, {"default_": 1.0}, 0.4, HW // 32, HW // 32)          # inference_lora.py:156, :247 at 512^2
    pctl = pc.AttentionReplace(*args, device=dev)
    revise_regionally_controlnet_forward(unet, pctl)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    concept = ConceptModels(unet, None)                                     # "1 concept, no LoRA": a concept pipe without adapters

    def hip(stage):
        pctl.reset()
        traj = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.generate_many([req], height=HW, width=HW, num_inference_steps=S, guidance_scale=gs, cross_attention_kwargs={"scale": 0.8},
                           controller=pctl, concept_models=concept, stage=stage, lora_list=["concept0"], styleL=False, trajectory=traj)
        torch.cuda.synchronize()
        return [t[0].float().cpu() for t in traj], time.perf_counter() - t0

    hip1, _ = hip(1)
    hip2, hip_s = hip(2)
    assert len(hip1) == len(hip2) == S
    # i > 15 never fires in a 10-step call (lora_pipeline.py:568): stage 2 is stage 1, bit for bit
    assert all(torch.equal(a, b) for a, b in zip(hip1, hip2))
    assert (pctl.cur_step, pctl.cur_att_layer) == (S, 0)

    # ---- the oracle on the same fp16-rounded weights and inputs
    f32 = lambda t: t.detach().float().cpu()
    sd = {k: f32(v) for k, v in unet.state_dict().items() if k in ou.param_shapes(ocfg)}
    assert len(sd) == len(ou.param_shapes(ocfg))
    ctx4 = torch.cat([f32(req["negative_prompt_embeds"]), f32(req["prompt_embeds"])])
    te4 = torch.cat([f32(req["negative_pooled_prompt_embeds"]), f32(req["pooled_prompt_embeds"])])
    tid = torch.tensor([[float(HW), float(HW), 0, 0, float(HW), float(HW)]]).repeat(4, 1)

    def oracle(stage, emulate):
        osch = osched.make("ddim", S)
        octl = oc.AttentionReplaceOracle(*args)
        octl.num_att_layers = pctl.num_att_layers
        attn = oc.reference_attn_fn(octl)
        rec = []

        def main(x, i):
            return ou.unet_forward(sd, ocfg, x, float(osch.timesteps[i]), ctx4, te4, tid, attn_fn=attn)

        t0 = time.perf_counter()
        with torch.no_grad():
            if emulate:
                with oprec.rounding(torch.float16):
                    opipe.denoise(main, [None], osch, req["latents"].float() * osch.init_noise_sigma, S, gs, stage, masks=[mask], record=rec)
            else:
                opipe.denoise(main, [None], osch, req["latents"].float() * osch.init_noise_sigma, S, gs, stage, masks=[mask], record=rec)
        return rec, time.perf_counter() - t0

    ref, ref_s = oracle(2, False)                  # the stage-2 call, timed for real: the CPU leg of configs[0]
    ref16, ref16_s = oracle(2, True)

    def curve(a, b):
        out = []
        for x, y, r in zip(a, b, ref):
            rms = r.pow(2).mean().sqrt().item()
            d = x - y
            out.append({"max": d.abs().max().item() / rms, "rms": d.pow(2).mean().sqrt().item() / rms})
        return out

    c_hip, c_emu, c_hip_emu = curve(hip2, ref), curve(ref16, ref), curve(hip2, ref16)
    n_fwd = 4 * S
    res = {"what": "BASELINE configs[0] end to end at FULL WIDTH: SDXL-base UNet (2,567,463,684 parameters, seeded random, fp16-rounded), 512x512 (latent 64x64), "
                   "10 DDIM steps, guidance 7.5, p2p controller installed (140 attention layers, self-replace for the first 4 steps), 1 concept without LoRA, "
                   "stage-2 call with a region mask (the fusion branch i > 15 never fires: stage 2 == stage 1, asserted bitwise on the HIP path); per-step error of the "
                   "latents (2, 4, 64, 64) over the fp32 oracle's latent rms of that step.  Reference call: lora_pipeline.py:485-632 with num_inference_steps=10",
           "steps": S, "sample_forwards_per_stage": n_fwd, "tflop_per_sample_forward": 1.590,
           "hip_vs_fp32_oracle": c_hip, "fp16_oracle_vs_fp32_oracle": c_emu, "hip_vs_fp16_oracle": c_hip_emu,
           "final": {"hip_vs_fp32_oracle": c_hip[-1], "fp16_oracle_vs_fp32_oracle": c_emu[-1], "hip_vs_fp16_oracle": c_hip_emu[-1]},
           "worst_step": {"hip_vs_fp32_oracle": {"max": max(c["max"] for c in c_hip), "rms": max(c["rms"] for c in c_hip)},
                          "fp16_oracle_vs_fp32_oracle": {"max": max(c["max"] for c in c_emu), "rms": max(c["rms"] for c in c_emu)}},
           "stage2_equals_stage1_bitwise_on_hip": True,
           "cpu_baseline_config0": {"wall_seconds_one_stage": ref_s, "images_per_sec_one_stage": 1.0 / ref_s, "wall_seconds_fp16_emulation": ref16_s,
                                    "tflops": n_fwd * 1.590 / ref_s, "threads": torch.get_num_threads(), "cpu_model": _cpu_model(), "torch": torch.__version__,
                                    "kind": "port", "what": "oracle/pipeline.denoise + oracle/unet.py + oracle/controller.py, fp32 torch on the GPU box's host cores: the REAL "
                                                            "10-step stage-2 call of configs[0] (40 sample-forwards), wall time, no extrapolation"},
           "hip_wall_seconds_one_stage_eager_single_request": hip_s}
    print("configs[0] full-width loop:", json.dumps(res))
    try:
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r06")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "config0_fullwidth_loop.json"), "w") as f:
            json.dump(res, f, indent=1)
    except OSError:
        pass
    # ten steps of the forward error (rms 1.1e-3 of an O(1) prediction, CFG 7.5 amplifies the conditional-unconditional difference) through the DDIM
    # update: a logic error (row order, controller window, scheduler table) is O(1).  And the HIP path must stay within twice the distance the
    # reference's own fp16 arithmetic keeps from exact arithmetic, at every step.
    for k in range(S):
        assert c_hip[k]["rms"] < 2.0 * c_emu[k]["rms"] + 2e-3, (k, c_hip[k], c_emu[k])
    assert c_hip[-1]["rms"] < 1.5e-2 and c_hip[-1]["max"] < 1e-1, c_hip[-1]
