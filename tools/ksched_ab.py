"""Within-probe A/B of K-loop schedule variants of the 256x256 four-wave GEMM: interleaved rounds in ONE process (cdna guide §5.4 rule 24),
median and best TF/s per variant, torch.equal against the shipped schedule.   python tools/ksched_ab.py 25,27,28 [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L
lib = L.lib(); dev = torch.device("cuda:0")
VARS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [25, 27, 28]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
def t(fn, n=6):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
SEL = sys.argv[3] if len(sys.argv) > 3 else "k"
shapes = [(65536, 10240, 1280, "geglu"), (65536, 1280, 1280, ""), (65536, 1280, 1280, "res"), (65536, 3840, 1280, ""), (65536, 1280, 5120, "res"), (262144, 640, 640, ""),
          (262144, 5120, 640, "geglu"), (8192, 8192, 8192, "")]
if SEL == "n640":      # the 640-wide Linear layers: 256-wide tiles pad N to 768
    shapes = [(262144, 640, 640, "res"), (262144, 640, 640, ""), (262144, 640, 2560, "res"), (131072, 640, 640, "res"), (262144, 1920, 640, ""), (262144, 5120, 640, "geglu"),
              (131072, 640, 2560, "res"), (65536, 1280, 1280, "res"), (32768, 1280, 1280, "res")]
for M, N, K, kind in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    b = torch.randn(N, device=dev, dtype=torch.float16)
    res = torch.randn(M, N, device=dev, dtype=torch.float16) if kind == "res" else None
    out = torch.empty(M, N // 2 if kind == "geglu" else N, device=dev, dtype=torch.float16)
    f = lambda: ops.gemm(x, w, bias=b, residual=res, act=L.ACT_GEGLU if kind == "geglu" else 0, out=out)
    ts, outs = {v: [] for v in VARS}, {}
    for r in range(R):
        for v in VARS:
            lib.omg_debug_set_gemm_variant(v)
            ts[v].append(t(f))
            if r == 0:
                outs[v] = out.clone()
    lib.omg_debug_set_gemm_variant(0)
    fl = 2 * M * N * K / 1e9
    med = lambda a: sorted(a)[len(a) // 2]
    print(f"{M}x{N}x{K} {kind:6s} " + "  ".join(f"v{v}: med {fl/med(ts[v]):6.0f} best {fl/min(ts[v]):6.0f} eq={int(torch.equal(outs[v], outs[VARS[0]]))}" for v in VARS), flush=True)
