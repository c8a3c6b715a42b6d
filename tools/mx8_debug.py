"""Probes for the MX-fp8 GEMM's data path and scale plumbing (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops
from oracle import mx8
dev = torch.device("cuda:0")
torch.manual_seed(0)

def run(a, w, tag):
    ta, tw = ops.quant_mx8(a.to(dev)), ops.quant_mx8(w.to(dev))
    qa, pa, ea = mx8.quantize(a.float()); qw, pw, ew = mx8.quantize(w.float())
    okq = torch.equal(ta.q.cpu(), qa) and torch.equal(tw.q.cpu(), qw)
    ref = mx8.dequantize(qa, ea) @ mx8.dequantize(qw, ew).T
    out = ops.gemm_mx8(ta, tw).float().cpu()
    err = (out - ref).abs()
    bad = err > 2e-3 + 2e-3 * ref.abs()
    print(f"{tag}: quant ok {okq}; bad {int(bad.sum())}/{bad.numel()} max err {err.max().item():.3e}; out[0,:4] {out[0,:4].tolist()} ref[0,:4] {ref[0,:4].tolist()}")
    if bad.any():
        rows = bad.any(dim=1).nonzero().flatten()[:8].tolist(); cols = bad.any(dim=0).nonzero().flatten()[:8].tolist()
        print("   bad rows", rows, "bad cols", cols, " ratio out/ref at first bad:", (out[bad][0] / ref[bad][0]).item() if ref[bad][0] != 0 else None)
    return out, ref

for K in (128, 256, 384):
    M = N = 256
    # (a) uniform scales: every 32-block has amax exactly 1
    a = (torch.rand(M, K) * 2 - 1) * 0.9; a.view(M, K // 32, 32)[:, :, 0] = 1.0
    w = (torch.rand(N, K) * 2 - 1) * 0.9; w.view(N, K // 32, 32)[:, :, 0] = -1.0
    run(a.half(), w.half(), f"K={K} (a) uniform scales, random data")
    # (b) identical data, per-block scales on A
    a = torch.ones(M, K); a.view(M, K // 32, 32).mul_((2.0 ** torch.arange(K // 32)).view(1, -1, 1) * (1 + torch.arange(M) % 3).view(-1, 1, 1).float().exp2())
    w = torch.ones(N, K)
    run(a.half(), w.half(), f"K={K} (b) identical data, A scales vary per block/row")
    # (b2) scales vary on W
    run(w.half(), a.half(), f"K={K} (b2) identical data, W scales vary")
    # (c) random data + scales
    a = torch.randn(M, K); a[:, 5] *= 30
    w = torch.randn(N, K) * K ** -0.5
    run(a.half(), w.half(), f"K={K} (c) random")
# ragged
a = torch.randn(300, 384); w = torch.randn(136, 384) * 0.05
run(a.half(), w.half(), "ragged 300x136x384")
