"""Run one GEMM shape a few times (for rocprofv3 --pmc runs).  python tools/gemm_one.py M N K variant iters [geglu]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L
M, N, K, v, it = (int(a) for a in sys.argv[1:6])
geglu = len(sys.argv) > 6
L.lib().omg_debug_set_gemm_variant(v)
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev, dtype=torch.float16)
w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.float16)
for _ in range(it):
    ops.gemm(x, w, out=out, act=L.ACT_GEGLU if geglu else L.ACT_NONE)
torch.cuda.synchronize()
