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
VARS = tuple(int(a, 0) for a in sys.argv[1:]) or (0,)      # EXP builds: 9 | us << 8 = variant 9 (= 8 + knobs) with half of the first-round workgroups started `us` late; | 1 << 16 = XCD-aware block order
print("# variants", VARS, "(0 = the heuristic: v3 above 128 keys, v2 below; 2 = v2; 3 = v3; 7 = EXP builds: v3 reading V row-major; 8 = 7 + three-address asm first MFMA) — OMG_HIP_LIB=<other build> runs the same table on another library for A/B")
print("# (B, heads, Nq, Nkv): TF/s per variant; max |last - first variant|")
for (B, heads, Nq, Nkv) in [(64, 10, 4096, 4096), (64, 20, 1024, 1024), (64, 20, 1024, 77), (64, 10, 4096, 77), (32, 10, 4096, 4096), (64, 20, 1024, 16)]:
    C = heads * 64
    q = torch.randn(B, Nq, C, device=dev, dtype=dt)
    k = torch.randn(B, Nkv, C, device=dev, dtype=dt) * 1.5
    v = torch.randn(B, Nkv, C, device=dev, dtype=dt)
    vt = ops.transpose_v(v, heads)
    if hasattr(lib, "omg_debug_set_attn_v"):          # EXP builds: variant 7 (tools/exp/attn_v7.h) reads V row-major, no transpose_v
        import ctypes
        lib.omg_debug_set_attn_v.argtypes, lib.omg_debug_set_attn_v.restype = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64], None
        lib.omg_debug_set_attn_v(v.data_ptr(), v.stride(1), v.stride(0))
    t_tr = timeit(lambda: ops.transpose_v(v, heads, out=vt)) if any(v & 0xff in (7, 8, 9) for v in VARS) and Nkv > 128 else None
    out = torch.empty(B, Nq, C, device=dev, dtype=dt)
    fl = 4.0 * B * heads * Nq * Nkv * 64
    res, outs = [], []
    for var in VARS:
        lib.omg_debug_set_attn_variant(var)
        ms = timeit(lambda: ops.attention(q, k, vt, heads, 0.125, out=out))
        res.append(fl / ms / 1e9)
        outs.append(out.clone())
    lib.omg_debug_set_attn_variant(0)
    print(f"({B},{heads},{Nq},{Nkv}): " + " | ".join(f"{r:7.0f}" for r in res) + f"   max |last - first| {(outs[0].float() - outs[-1].float()).abs().max().item():.2e}"
          + (f"   transpose_v {t_tr * 1e3:.0f} us (the pass variant 7 makes unnecessary; attention itself {fl / res[0] / 1e6:.0f} us)" if t_tr is not None else ""))
