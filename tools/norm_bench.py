"""GroupNorm / LayerNorm at the benchmark's batch (64 samples per fused step).  python tools/norm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops
dev, dt = torch.device("cuda:0"), torch.float16
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
R = lambda *s: torch.randn(*s, device=dev, dtype=dt)
for (B, HW, C, C2) in [(64, 16384, 320, 0), (64, 4096, 640, 0), (64, 1024, 1280, 0), (64, 1024, 1280, 1280), (64, 4096, 640, 320), (64, 16384, 320, 320), (32, 16384, 320, 0)]:
    x = R(B, HW, C); x2 = R(B, HW, C2) if C2 else None
    g, b = R(C + C2), R(C + C2)
    ms = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, silu=True, x2=x2))
    n = B * HW * (C + C2)
    print(f"gn  B={B} HW={HW:6d} C={C}+{C2}: {ms*1e3:8.1f} us  {3*n*2/ms/1e6:8.0f} GB/s (2R+1W)")
for (M, C) in [(262144, 640), (65536, 1280), (32768, 1280)]:
    x = R(M, C); g, b = R(C), R(C)
    ms = timeit(lambda: ops.layernorm(x, g, b, 1e-5))
    print(f"ln  M={M} C={C}: {ms*1e3:8.1f} us  {2*x.numel()*2/ms/1e6:8.0f} GB/s (1R+1W)")
    ms = timeit(lambda: ops.layernorm_mx8(x, g, b, 1e-5))
    print(f"ln->mx8 M={M} C={C}: {ms*1e3:8.1f} us  {x.numel()*3/ms/1e6:8.0f} GB/s (2 B read + 1 B written)")
