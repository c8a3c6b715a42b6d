"""Run one attention shape a few times (for rocprofv3 --pmc runs).  python tools/attn_one.py B heads Nq Nkv variant iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L
B, heads, Nq, Nkv, v, it = (int(a) for a in sys.argv[1:7])
L.lib().omg_debug_set_attn_variant(v)
dev = torch.device("cuda:0")
C = heads * 64
q = torch.randn(B, Nq, C, device=dev, dtype=torch.float16)
k = torch.randn(B, Nkv, C, device=dev, dtype=torch.float16) * 1.5
vv = torch.randn(B, Nkv, C, device=dev, dtype=torch.float16)
vt = ops.value_operand(vv, heads)          # above 128 keys: the row-major view itself (attn_fwd_kernel7), else the V^T image
out = torch.empty(B, Nq, C, device=dev, dtype=torch.float16)
for _ in range(it):
    ops.attention(q, k, vt, heads, 0.125, out=out)
torch.cuda.synchronize()
