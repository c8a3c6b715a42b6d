"""Do two independent lock-step engines on two HIP streams beat one engine of twice the batch?  (round 6, late.)

The benchmark step is 8 requests batched through one engine: every kernel of a step runs alone on the device, so the HBM-bound kernels between the
GEMMs (LayerNorm / GroupNorm / elementwise: ~6.5 % of the step) leave the matrix pipes idle and the GEMMs leave HBM idle.  Two engines of 4 requests
each — own UNet object, controller, graphs and buffers; the same seeded weights — replayed from two host threads on two streams let the hardware
co-schedule one engine's small kernels under the other's GEMMs (a persistent GEMM block leaves 128 registers per SIMD lane free; LDS-free kernels fit).
Measured here: (a) one engine x 8 requests, (b) two engines x 4 one after the other, (c) two engines x 4 concurrently.   python tools/two_streams_probe.py [--steps 2]"""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from omg_amd import controller as pc
from omg_amd.pipeline import LoraMultiConceptPipeline, revise_regionally_controlnet_forward
from omg_amd.schedulers import make_scheduler
from omg_amd.synthetic import c2_inputs, c2_masks, make_concept_models
from omg_amd.unet import UNet2DConditionModel, UNetConfig
from omg_amd.vae import AutoencoderKLDecoder, VaeConfig

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--tiny", action="store_true")
ap.add_argument("--denoise-steps", type=int, default=50)
a = ap.parse_args()
dev, dt = torch.device("cuda:0"), torch.float16
cfg = UNetConfig.tiny() if a.tiny else UNetConfig.sdxl()
HW = cfg.sample_size * 8
P = "a man and a woman walking on the street"


class Setup:
    def __init__(self):
        self.unet = UNet2DConditionModel(cfg, dtype=dt, device=dev).init_synthetic_(seed=0)
        self.ctl = pc.AttentionReplace([P, P], 50, cross_replace_steps={"default_": 1.0}, self_replace_steps=0.4, width=HW // 32, height=HW // 32, device=dev, dtype=dt)
        B._quiet(revise_regionally_controlnet_forward, self.unet, self.ctl)
        self.concept = make_concept_models(self.unet, n_concepts=2, rank=8 if a.tiny else 64)
        self.pipe = LoraMultiConceptPipeline(self.unet, make_scheduler("ddim"))
        self.vae = AutoencoderKLDecoder(VaeConfig.tiny() if a.tiny else VaeConfig.sdxl(), dtype=dt, device=dev, upcast=True).init_synthetic_(seed=1)
        self.stream = torch.cuda.Stream(device=dev)

    def run(self, reqs):
        self.ctl.reset()
        lat = self.pipe.generate_many(reqs, height=HW, width=HW, num_inference_steps=a.denoise_steps, guidance_scale=7.5, cross_attention_kwargs={"scale": 0.8},
                                      controller=self.ctl, concept_models=self.concept, stage=2, lora_list=["concept0", "concept1"], styleL=False, use_graph=True)
        for j in range(lat.shape[0]):
            img = self.vae.decode_latents(lat[j])
        return lat


masks = c2_masks(HW, HW, device=dev)
S = [Setup(), Setup()]


def reqs(seed0, n):
    out = []
    for j in range(n):
        r = c2_inputs(S[0].unet, seed=seed0 * 16 + j, height=HW, width=HW)
        r["region_masks"] = masks
        out.append(r)
    return out


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


r8 = reqs(1, 8)
# warm-up: every engine captures its graphs alone (a capture must not see another thread's launches)
with torch.cuda.stream(S[0].stream):
    S[0].run(r8)
torch.cuda.synchronize()
for s, rr in zip(S, (r8[:4], r8[4:])):
    with torch.cuda.stream(s.stream):
        s.run(rr)
    torch.cuda.synchronize()


def one_engine():
    with torch.cuda.stream(S[0].stream):
        return S[0].run(r8)


def sequential():
    out = []
    for s, rr in zip(S, (r8[:4], r8[4:])):
        with torch.cuda.stream(s.stream):
            out.append(s.run(rr))
    return torch.cat(out)


def concurrent():
    out = [None, None]

    def work(i, rr):
        with torch.cuda.stream(S[i].stream):
            out[i] = S[i].run(rr)
            S[i].stream.synchronize()

    th = [threading.Thread(target=work, args=(i, rr)) for i, rr in enumerate((r8[:4], r8[4:]))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return torch.cat(out)


for rep in range(a.steps):
    t1, l1 = timed(one_engine)
    t2, l2 = timed(sequential)
    t3, l3 = timed(concurrent)
    print(f"round {rep}: one engine x 8 requests {t1:.3f} s ({8 / t1:.4f} img/s) | two engines x 4, one after the other {t2:.3f} s ({8 / t2:.4f}) | "
          f"two engines x 4 on two streams {t3:.3f} s ({8 / t3:.4f} img/s, {100 * (t1 / t3 - 1):+.1f} % vs one engine); latents equal: seq {bool(torch.equal(l1, l2))} conc {bool(torch.equal(l1, l3))}", flush=True)
