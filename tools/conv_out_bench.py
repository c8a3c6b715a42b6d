"""conv_out (the UNet's last convolution, 320 -> 4 channels) at the benchmark's shapes: time and effective input bandwidth.  python tools/conv_out_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from omg_amd import ops
dev = torch.device("cuda:0")
for (B, H, W, Cin, Cout) in [(64, 128, 128, 320, 4), (8, 128, 128, 320, 4), (2, 24, 20, 64, 4)]:
    x = torch.randn(B, H, W, Cin, device=dev, dtype=torch.float16)
    w = torch.randn(Cout, 3, 3, Cin, device=dev, dtype=torch.float16) * (9 * Cin) ** -0.5
    b = torch.randn(Cout, device=dev, dtype=torch.float16)
    y = ops.conv_out(x, w, b)
    if B <= 8:
        ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), b.float().cpu(), padding=1)
        err = (y.cpu() - ref).abs().max().item()
    else:
        err = float("nan")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.conv_out(x, w, b, out=y)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"conv_out {B}x{H}x{W}x{Cin} -> {Cout}: {us:8.1f} us  ({x.numel() * 2 / us / 1e6:.2f} TB/s of input)   max |d| vs F.conv2d {err:.2e}")
