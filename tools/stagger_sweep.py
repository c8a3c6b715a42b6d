"""De-phasing the CUs (debug bits 16..23 of the GEMM variant word = S in 0.25 us): TF/s of the benchmark's GEMM shapes against S.
python tools/stagger_sweep.py [S values, comma separated]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L
lib = L.lib(); dev = torch.device("cuda:0")
SV = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 4, 8, 16, 32, 64, 128]
def t(fn, n=8):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
shapes = [(65536, 10240, 1280, "geglu"), (65536, 1280, 1280, "res"), (65536, 3840, 1280, ""), (65536, 1280, 5120, "res"), (262144, 5120, 640, "geglu"),
          (262144, 640, 640, "res"), (262144, 1920, 640, "")]
for M, N, K, kind in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    b = torch.randn(N, device=dev, dtype=torch.float16)
    res = torch.randn(M, N, device=dev, dtype=torch.float16) if kind == "res" else None
    out = torch.empty(M, N // 2 if kind == "geglu" else N, device=dev, dtype=torch.float16)
    f = lambda: ops.gemm(x, w, bias=b, residual=res, act=L.ACT_GEGLU if kind == "geglu" else 0, out=out)
    row = []
    for rnd in range(2):
        for S in SV:
            lib.omg_debug_set_gemm_variant(25 | ((S << 16) << 8))
            row.append((rnd, S, 2 * M * N * K / t(f) / 1e9))
    lib.omg_debug_set_gemm_variant(0)
    best = {}
    for rnd, S, v in row:
        best[S] = max(best.get(S, 0), v)
    print(f"{M}x{N}x{K} {kind:6s} " + "  ".join(f"S={S*0.25:5.2f}us:{best[S]:6.0f}" for S in SV), flush=True)
