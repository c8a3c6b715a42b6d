"""Which scale does the hardware apply to the byte at K position k0?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops
dev = torch.device("cuda:0")
M = N = 256; K = 128
def scales(rows, exps):   # exps: 4 ints per stage -> packed dword for every row
    d = sum(((127 + e) & 0xff) << (8 * i) for i, e in enumerate(exps))
    if d >= 2**31: d -= 2**32
    return torch.full((1, rows), d, dtype=torch.int32, device=dev)
ones = torch.full((N, K), 0x38, dtype=torch.uint8, device=dev)
for side in ("A", "W"):
    res = []
    for k0 in range(K):
        q = torch.zeros((M, K), dtype=torch.uint8, device=dev); q[:, k0] = 0x38
        varying = ops.Mx8Tensor(q, scales(M, [0, 1, 2, 3]))
        flat = ops.Mx8Tensor(ones, scales(N, [0, 0, 0, 0]))
        out = ops.gemm_mx8(varying, flat) if side == "A" else ops.gemm_mx8(flat, varying)
        v = out[3, 5].item()
        res.append(int(round(torch.log2(torch.tensor(v)).item())) if v > 0 else -1)
    print(side, "scale block applied to byte k0:", "".join(str(r) for r in res))
