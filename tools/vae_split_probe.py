"""What a split-bf16 VAE up block would cost, MEASURED on the decoder's own convolution shapes (GPU box only):

    python tools/vae_split_probe.py

x = hi + mid + lo (three bf16 terms = 24 significant bits) turns one exact-fp32 convolution into SIX bf16 products with fp32 accumulation
(hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid), i.e. an implicit GEMM with six times the K on the bf16 MFMA; a two-term split (16 bits,
hi.hi, hi.lo, lo.hi) is three times the K.  This probe times, for every 3x3 convolution shape of the SDXL decoder's fp32 up blocks (two
1024^2 images per call), the shipped exact kernel (omg_conv2d_f32 on v_mfma_f32_32x32x2_f32) and the shipped bf16 implicit-GEMM convolution
(omg_conv2d, gemm_kernel_v12 / v13 / v7) on an input with 6x / 3x the channels — the arithmetic a split kernel would execute, WITHOUT the cost of
writing the [hi | mid | lo] feature maps (GroupNorm + SiLU would have to) and of an fp32-output epilogue: a lower bound on its time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops

dev = torch.device("cuda:0")
# (H, W of the conv's output, Cin, Cout, count per decode of the up blocks + conv_norm_out path); B = 2 images
SHAPES = [(128, 128, 512, 512, 6), (256, 256, 512, 512, 7), (512, 512, 512, 256, 1), (512, 512, 256, 256, 6), (1024, 1024, 256, 128, 1), (1024, 1024, 128, 128, 6)]


def t(fn, n=4):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


tot = {"f32": 0.0, "x6": 0.0, "x3": 0.0}
print("# shape (B=2): exact fp32 kernel | bf16 conv at 6x K | at 3x K   (ms per launch; TF/s of the fp32-equivalent FLOPs)")
for H, W, Ci, Co, cnt in SHAPES:
    B = 2
    fl = 2.0 * B * H * W * Co * 9 * Ci
    x32 = torch.randn(B, H, W, Ci, device=dev)
    w32 = torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5
    ms32 = t(lambda: ops.conv2d_f32(x32, w32, 3))
    row = [ms32]
    for mult in (6, 3):
        xb = torch.randn(B, H, W, Ci * mult, device=dev, dtype=torch.bfloat16)
        wb = (torch.randn(Co, 9 * Ci * mult, device=dev) * (9 * Ci * mult) ** -0.5).to(torch.bfloat16)
        row.append(t(lambda: ops.conv2d(xb, wb, 3)))
        del xb, wb
    tot["f32"] += cnt * row[0]; tot["x6"] += cnt * row[1]; tot["x3"] += cnt * row[2]
    print(f"{H:5d}x{W:<5d} {Ci:4d}->{Co:<4d} x{cnt}: {row[0]:7.2f} ms ({fl / row[0] / 1e9:5.0f} TF/s) | {row[1]:7.2f} ms ({fl / row[1] / 1e9:5.0f}) | {row[2]:7.2f} ms ({fl / row[2] / 1e9:5.0f})", flush=True)
print(f"sum over the up blocks' 3x3 convolutions of one decode call (2 images): exact fp32 {tot['f32']:.1f} ms | 3-term split (6 products) >= {tot['x6']:.1f} ms | "
      f"2-term split (3 products) >= {tot['x3']:.1f} ms")
