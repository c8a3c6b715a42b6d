import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from omg_amd import ops, _lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, C = 300, 640
N = 8 * C
a = torch.randn(M, C, device=dev, dtype=torch.float16)
w = (torch.randn(N, C, device=dev) * C ** -0.5).half()
b = torch.randn(N, device=dev, dtype=torch.float16)
perm = ops.geglu_row_perm(N).to(dev)
wp, bp = w[perm].contiguous(), b[perm].contiguous()
h = a.float() @ w.float().T + b.float()
val, gate = h.chunk(2, dim=-1)
ref = val * F.gelu(gate)
for v in (13, 15, 15, 15):
    L.lib().omg_debug_set_gemm_variant(v)
    out = ops.gemm(a, wp, bias=bp, act=L.ACT_GEGLU).float()
    bad = ~torch.isclose(out, ref, rtol=4e-3, atol=8e-3)
    idx = bad.nonzero()
    print("variant", v, "bad", int(bad.sum()), "nan", int(torch.isnan(out).sum()))
    if len(idx):
        rows, cols = idx[:, 0], idx[:, 1]
        print("  rows", sorted(set((rows // 32).tolist())), "(32-blocks)  cols 16-blocks", sorted(set((cols // 16).tolist()))[:40])
        print("  first", idx[:5].tolist(), out[idx[0, 0], idx[0, 1]].item(), ref[idx[0, 0], idx[0, 1]].item())
    # plain gemm for the same inputs
    o2 = ops.gemm(a, wp, bias=bp).float()
    r2 = a.float() @ wp.float().T + bp.float()
    bad2 = ~torch.isclose(o2, r2, rtol=4e-3, atol=8e-3)
    print("  plain bad", int(bad2.sum()))
