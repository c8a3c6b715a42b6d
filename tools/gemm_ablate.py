"""Ablation of the v7 K loop (fp16 plain GEMM): which of LDS-DMA issue / fragment reads / the stage barrier costs what.
Variants 17..23 = v7 with ablation bits (1 = no DMA in the loop, 2 = no fragment reads, 4 = no wait+barrier); results of the
ablated kernels are wrong by construction, only their time is meaningful.  Needs the library built with `make -C omg_amd/csrc ABLATE=1`
(the ablation instantiations are left out of the default build).  python tools/gemm_ablate.py [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 8192, 8192)
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev, dtype=torch.float16)
w = torch.randn(N, K, device=dev, dtype=torch.float16)
out = torch.empty(M, N, device=dev, dtype=torch.float16)


def timeit(fn, iters=10, warm=3):
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


names = {0: "full", 1: "no DMA", 2: "no frag reads", 3: "no DMA, no reads", 4: "no barrier", 5: "no DMA, no barrier", 6: "no reads, no barrier", 7: "MFMA only"}
for rep in range(2):
    for abl in range(8):
        L.lib().omg_debug_set_gemm_variant(15 if abl == 0 else 16 + abl)
        ms = timeit(lambda: ops.gemm(x, w, out=out))
        print(f"{names[abl]:22s}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.0f} TF/s")
L.lib().omg_debug_set_gemm_variant(0)
