"""Epilogue cost of the large-tile GEMM kernels: the same shape timed with and without the epilogue (debug bit 32),
plain / bias+residual / GEGLU, for the 4-wave 256x256 (15), its persistent form (16), the 8-wave 256x256 (13) and 256x128 (14)
kernels; plus the device's HBM write / copy rates for scale.  python tools/gemm_epi.py [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 10240, 1280)
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev, dtype=torch.float16)
w = torch.randn(N, K, device=dev, dtype=torch.float16)
b = torch.randn(N, device=dev, dtype=torch.float16)
res = torch.randn(M, N, device=dev, dtype=torch.float16)


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for v in (13, 14, 15, 16):
    for name, kw in (("plain", {}), ("bias+res", dict(bias=b, residual=res)), ("geglu", dict(bias=b, act=L.ACT_GEGLU))):
        row = []
        for dbg in (0, 32):
            L.lib().omg_debug_set_gemm_variant(v | (dbg << 8))
            row.append(timeit(lambda: ops.gemm(x, w, **kw)))
        print(f"variant {v} {name:9s}: {row[0]*1e3:8.1f} us ({2*M*N*K/row[0]/1e9:6.0f} TF/s)   without epilogue {row[1]*1e3:8.1f} us ({2*M*N*K/row[1]/1e9:6.0f} TF/s)")
L.lib().omg_debug_set_gemm_variant(0)
buf = torch.empty(M * N, device=dev, dtype=torch.float16)
src = torch.randn(M * N, device=dev, dtype=torch.float16)
ms = timeit(lambda: buf.fill_(1.0))
print(f"fill_ {buf.numel()*2/1e6:.0f} MB: {ms*1e3:.1f} us = {buf.numel()*2/ms/1e9:.2f} TB/s write")
ms = timeit(lambda: buf.copy_(src))
print(f"copy_ {buf.numel()*2/1e6:.0f} MB: {ms*1e3:.1f} us = {2*buf.numel()*2/ms/1e9:.2f} TB/s read+write")
