"""Run one MX-fp8 GEMM shape a few times (for rocprofv3 --pmc runs).  python tools/mx8_one.py M N K iters [geglu]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L
M, N, K, it = (int(a) for a in sys.argv[1:5])
geglu = len(sys.argv) > 5
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev, dtype=torch.float16)
w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
xq, wq = ops.quant_mx8(x), ops.quant_mx8(w)
out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.float16)
for _ in range(it):
    ops.gemm_mx8(xq, wq, out=out, act=L.ACT_GEGLU if geglu else L.ACT_NONE)
torch.cuda.synchronize()
