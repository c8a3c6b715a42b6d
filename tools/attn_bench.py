"""Flash-attention kernel variants on the benchmark's shapes (GPU box).  python tools/attn_bench.py [v v ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L


def timeit(fn, iters=30, warm=10):
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


dev, dt = torch.device("cuda:0"), torch.float16
lib = L.lib()
VARS = tuple(int(a, 0) for a in sys.argv[1:]) or (0,)
print("# variants", VARS, "(0 = the heuristic: v7 above 128 keys — V row-major, no transpose_v — v6 / v2 below; 2 = v2 on the V^T image (+ the transpose_v pass it needs, timed separately); 6 = v6; 7 = v7)"
      " — OMG_HIP_LIB=<other build> runs the same table on another library for A/B")
print("# q, k, v are the three column slices of ONE fused (B, N, 3C) projection output, as the UNet has them")
print("# (B, heads, Nq, Nkv): TF/s per variant; max |last - first variant|")
for (B, heads, Nq, Nkv) in [(64, 10, 4096, 4096), (64, 20, 1024, 1024), (64, 20, 1024, 77), (64, 10, 4096, 77), (32, 10, 4096, 4096), (64, 20, 1024, 16)]:
    C = heads * 64
    if Nq == Nkv:
        qkv = torch.randn(B, Nq, 3 * C, device=dev, dtype=dt)
        qkv[:, :, C:2 * C] *= 1.5
        q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    else:
        q = torch.randn(B, Nq, C, device=dev, dtype=dt)
        k = torch.randn(B, Nkv, C, device=dev, dtype=dt) * 1.5
        v = torch.randn(B, Nkv, C, device=dev, dtype=dt)
    vt = ops.transpose_v(v, heads)
    vr = ops.value_operand(v, heads)               # RowMajorV above 128 keys, else the V^T image
    t_tr = timeit(lambda: ops.transpose_v(v, heads, out=vt)) if Nkv > 128 else None
    out = torch.empty(B, Nq, C, device=dev, dtype=dt)
    fl = 4.0 * B * heads * Nq * Nkv * 64
    res, outs = [], []
    for var in VARS:
        operand = vr if var in (0, 7) else vt
        lib.omg_debug_set_attn_variant(var)
        ms = timeit(lambda: ops.attention(q, k, operand, heads, 0.125, out=out))
        res.append(fl / ms / 1e9)
        outs.append(out.clone())
    lib.omg_debug_set_attn_variant(0)
    print(f"({B},{heads},{Nq},{Nkv}): " + " | ".join(f"{r:7.0f}" for r in res) + f"   max |last - first| {(outs[0].float() - outs[-1].float()).abs().max().item():.2e}"
          + (f"   transpose_v {t_tr * 1e3:.0f} us (what v3 needs on top; attention itself {fl / res[0] / 1e6:.0f} us)" if t_tr is not None else ""))
