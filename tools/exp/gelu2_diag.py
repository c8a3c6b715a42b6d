"""GELU2 experiment (tools/exp/gelu_v2.h): which tile variants disagree on a GEGLU launch, by how much, and on which gate values.
OMG_HIP_LIB=tools/exp/build/gelu2/libomg_hip.so python tools/exp/gelu2_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from omg_amd import ops, _lib as L
lib = L.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, N, K = 600, 704, 320
a = torch.randn(M, K, generator=g).half().to(dev)
w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
b = torch.randn(N, generator=g).half().to(dev)
perm = ops.geglu_row_perm(N).to(dev)
wg, bg = w[perm].contiguous(), b[perm].contiguous()
outs = {}
for v in (1, 13, 14, 15, 24, 25, 28):
    lib.omg_debug_set_gemm_variant(v)
    outs[v] = ops.gemm(a, wg, bias=bg, act=L.ACT_GEGLU).clone()
lib.omg_debug_set_gemm_variant(0)
full = (a.float() @ w.float().T + b.float())
val, gate = full[:, : N // 2], full[:, N // 2:]
for v, o in outs.items():
    d = (o.view(torch.int16).int() - outs[1].view(torch.int16).int())
    nz = d != 0
    print(f"variant {v}: {int(nz.sum())} of {o.numel()} outputs differ from variant 1; max |ulp| {int(d.abs().max())}", end="")
    if nz.any():
        idx = nz.nonzero()[:5]
        print("  e.g. gate values", [round(float(gate[i, j]), 4) for i, j in idx.tolist()], "outputs", [(float(o[i, j]), float(outs[1][i, j])) for i, j in idx.tolist()][:3])
    else:
        print()
