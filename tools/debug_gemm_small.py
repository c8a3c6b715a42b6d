import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in [(960, 320, 64), (960, 320, 128), (960, 384, 64), (1024, 320, 64)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev, dtype=torch.float16)
    ref = a.float() @ w.float().T + b.float()
    for v in (13, 14, 15, 13, 13):
        L.lib().omg_debug_set_gemm_variant(v)
        out = ops.gemm(a, w, bias=b).float()
        bad = ~torch.isclose(out, ref, rtol=4e-3, atol=8e-3)
        idx = bad.nonzero()
        print((M, N, K), "variant", v, "bad", int(bad.sum()))
        if len(idx):
            rows, cols = idx[:, 0], idx[:, 1]
            print("   rows", sorted(set(rows.tolist()))[:24], " cols", sorted(set(cols.tolist()))[:40])
            r, c = idx[0].tolist()
            print("   first", (r, c), out[r, c].item(), ref[r, c].item(), " bias", b[c].item(), " partial sums", [(a[r, :k].float() @ w[c, :k].float() + b[c].float()).item() for k in (16, 32, 48, 64)])
