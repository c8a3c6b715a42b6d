"""VAE decode (row N1) at the benchmark's image size: time and rate of AutoencoderKLDecoder.decode for the two 1024x1024
images of one stage-2 call (latents (2, 4, 128, 128)).  python tools/vae_bench.py [--dtype bf16|fp16] [--batch 2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops
from omg_amd.vae import AutoencoderKLDecoder, VaeConfig

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--latent", type=int, default=128)
ap.add_argument("--upcast", type=int, default=0, help="1: the reference's decode (fp16 mid block, fp32 up blocks)")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
dev = torch.device("cuda:0")
vae = AutoencoderKLDecoder(VaeConfig.sdxl(), dtype=dt, device=dev, upcast=bool(a.upcast)).init_synthetic_(0)
z = torch.randn(a.batch, 4, a.latent, a.latent, device=dev)
prof = ops.KernelProfiler()
img = vae.decode(z)                      # warm-up (weight packing, code objects)
torch.cuda.synchronize()
ops.set_profiler(prof)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
img = vae.decode(z)
e.record()
torch.cuda.synchronize()
ops.set_profiler(None)
ms = s.elapsed_time(e)
sm = prof.summary()
fl = sum(d["flops"] for d in sm.values())
print(f"decode {tuple(z.shape)} -> {tuple(img.shape)} {a.dtype} upcast={a.upcast}: {ms:.1f} ms (instrumented); " + "; ".join(
      f"{k} {d['flops']/1e12:.2f} TFLOP in {d['ms']:.1f} ms = {d['flops']/d['ms']/1e9:.0f} TF/s" for k, d in sm.items()) + f"; finite={bool(torch.isfinite(img).all())}")
for (kind, tag), d in prof.by_tag()[:8]:
    print(f"  {kind} {str(tag):44s} n={d['launches']:3d} ms={d['ms']:7.2f} {d['flops']/d['ms']/1e9:7.0f} TF/s")
s.record()
for _ in range(3):
    img = vae.decode(z)
e.record()
torch.cuda.synchronize()
print(f"un-instrumented: {s.elapsed_time(e)/3:.1f} ms per call of {a.batch} images")
