"""v7 (variant 15) against v7 + transposed streaming epilogue (variant 25) on the UNet's GEMM / conv shapes; checks torch.equal.
python tools/xe_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L
lib = L.lib(); dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
shapes = [(65536, 10240, 1280, "geglu"), (32768, 10240, 1280, "geglu"), (65536, 1280, 1280, "res"), (65536, 1280, 1280, ""), (65536, 1280, 5120, "res"), (65536, 3840, 1280, ""),
          (262144, 640, 640, "res"), (262144, 640, 640, ""), (262144, 5120, 640, "geglu"), (262144, 640, 2560, "res"), (262144, 1920, 640, ""), (32768, 1280, 1280, "res"),
          (32768, 1280, 5120, "res"), (32768, 3840, 1280, ""), (4928, 2560, 2048, "")]
for M, N, K, kind in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    b = torch.randn(N, device=dev, dtype=torch.float16)
    res = torch.randn(M, N, device=dev, dtype=torch.float16) if kind == "res" else None
    out = torch.empty(M, N // 2 if kind == "geglu" else N, device=dev, dtype=torch.float16)
    row, outs = [], []
    for v in (15, 25):
        lib.omg_debug_set_gemm_variant(v)
        ms = t(lambda: ops.gemm(x, w, bias=b, residual=res, act=L.ACT_GEGLU if kind == "geglu" else 0, out=out))
        outs.append(out.clone()); row.append(2 * M * N * K / ms / 1e9)
    lib.omg_debug_set_gemm_variant(0)
    print(f"lin  {M}x{N}x{K} {kind:6s} v7 {row[0]:7.0f} TF/s  v7+XE {row[1]:7.0f} TF/s  ({row[1]/row[0]-1:+.1%})  equal={torch.equal(outs[0], outs[1])}")
convs = [(8, 32, 32, 1280, 1280, True), (8, 32, 32, 2560, 1280, False), (8, 64, 64, 640, 640, True), (4, 128, 128, 320, 320, True), (8, 64, 64, 1280, 640, False)]
for B, H, W, Cin, Cout, with_res in convs:
    x = torch.randn(B, H, W, Cin, device=dev, dtype=torch.float16)
    w = torch.randn(Cout, 9 * Cin, device=dev, dtype=torch.float16) * (9 * Cin) ** -0.5
    b = torch.randn(Cout, device=dev, dtype=torch.float16)
    res = torch.randn(B, H, W, Cout, device=dev, dtype=torch.float16) if with_res else None
    row, outs = [], []
    for v in (15, 25):
        lib.omg_debug_set_gemm_variant(v)
        f = lambda: ops.conv2d(x, w, 3, bias=b, residual=res)
        ms = t(f); outs.append(f()); row.append(2 * B * H * W * Cout * 9 * Cin / ms / 1e9)
    lib.omg_debug_set_gemm_variant(0)
    print(f"conv {B}x{H}x{W} {Cin}->{Cout} res={with_res} v7 {row[0]:7.0f}  v7+XE {row[1]:7.0f} TF/s ({row[1]/row[0]-1:+.1%}) equal={torch.equal(outs[0], outs[1])}")
