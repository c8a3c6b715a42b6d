"""Within-probe A/B of K-loop schedule variants of the 256x256 four-wave GEMM: interleaved rounds in ONE process (cdna guide §5.4 rule 24),
median and best TF/s per variant, torch.equal against the first variant listed (0 = the heuristic's choice).
python tools/ksched_ab.py 25,27,28 [rounds] [k | n320 | n640 | slots | geglu | conv | conv320]"""
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
if SEL == "geglu":     # the GEGLU projections of the workload (tools/gpu_exp_gelu2.sh: one run per library)
    shapes = [(65536, 10240, 1280, "geglu"), (32768, 10240, 1280, "geglu"), (262144, 5120, 640, "geglu"), (131072, 5120, 640, "geglu")]
if SEL == "n320":      # every Linear width of the workload that is a whole number of 320-wide tiles (the 256x320 tile, variants 27 / 28), GEGLU excluded
    shapes = [(65536, 1280, 1280, "res"), (65536, 1280, 1280, ""), (65536, 3840, 1280, ""), (65536, 1280, 5120, "res"), (262144, 640, 640, "res"), (262144, 640, 640, ""),
              (262144, 1920, 640, ""), (262144, 640, 2560, "res"), (32768, 1280, 1280, "res"), (32768, 3840, 1280, "")]
if SEL == "n640":      # the 640-wide Linear layers: 256-wide tiles pad N to 768
    shapes = [(262144, 640, 640, "res"), (262144, 640, 640, ""), (262144, 640, 2560, "res"), (131072, 640, 640, "res"), (262144, 1920, 640, ""), (262144, 5120, 640, "geglu"),
              (131072, 640, 2560, "res"), (65536, 1280, 1280, "res"), (32768, 1280, 1280, "res")]
if SEL in ("conv", "conv320"):      # 3x3 convolutions of the UNet: (B, H, W, Cin, Cout).  conv: the ones on the 256x256 tile; conv320: the N = 320 / 640
    # ones the heuristic (variant 0) gives to the 128x320 tile — the 256x320 tile's (27 / 28, tools/exp/gemm_v13.h) first target
    CONVS = [(64, 32, 32, 1280, 1280), (64, 64, 64, 640, 640), (64, 32, 32, 2560, 1280), (64, 64, 64, 1280, 640), (16, 128, 128, 640, 640)]
    if SEL == "conv320":
        CONVS = [(64, 128, 128, 320, 320), (64, 128, 128, 640, 320), (64, 128, 128, 960, 320), (64, 64, 64, 640, 640), (64, 64, 64, 1280, 640), (64, 64, 64, 320, 640)]
    for B, H, W_, Ci, Co in CONVS:
        x = torch.randn(B, H, W_, Ci, device=dev, dtype=torch.float16)
        w = torch.randn(Co, 9 * Ci, device=dev, dtype=torch.float16) * (9 * Ci) ** -0.5
        b = torch.randn(Co, device=dev, dtype=torch.float16)
        gb = torch.randn(B, Co, device=dev, dtype=torch.float16)
        f = lambda: ops.conv2d(x, w, 3, bias=b, group_bias=gb)
        ts, outs = {v: [] for v in VARS}, {}
        for r in range(R):
            for v in VARS:
                lib.omg_debug_set_gemm_variant(v)
                ts[v].append(t(f))
                if r == 0:
                    outs[v] = f().clone()
        lib.omg_debug_set_gemm_variant(0)
        fl = 2 * B * H * W_ * Co * 9 * Ci / 1e9
        med = lambda a: sorted(a)[len(a) // 2]
        print(f"conv {B}x{H}x{W_} {Ci}->{Co} " + "  ".join(f"v{v}: med {fl/med(ts[v]):6.0f} best {fl/min(ts[v]):6.0f} eq={int(torch.equal(outs[v], outs[VARS[0]]))}" for v in VARS), flush=True)
    sys.exit(0)
if SEL == "slots":     # the transformer Linears as the fused steps launch them: 64 samples, each with its own weight slot (merged LoRA), adapter ids on the device
    shapes = [(65536, 3840, 1280, "slots"), (65536, 1280, 1280, "slots+res"), (65536, 1280, 5120, "slots+res"), (65536, 10240, 1280, "slots+geglu"), (262144, 640, 640, "slots+res")]
for M, N, K, kind in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    if kind.startswith("slots"):
        w3 = torch.randn(3, N, K, device=dev, dtype=torch.float16) * K ** -0.5
        ids = torch.tensor([(g * 7) % 3 for g in range(64)], dtype=torch.int32, device=dev)
        b = torch.randn(N, device=dev, dtype=torch.float16)
        res = torch.randn(M, N, device=dev, dtype=torch.float16) if "res" in kind else None
        geglu = "geglu" in kind
        out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.float16)
        f = lambda: ops.gemm(x, w3, bias=b, residual=res, act=L.ACT_GEGLU if geglu else 0, groups=64, w_group_adapter=ids, out=out)
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
        print(f"{M}x{N}x{K} {kind:12s} " + "  ".join(f"v{v}: med {fl/med(ts[v]):6.0f} best {fl/min(ts[v]):6.0f} eq={int(torch.equal(outs[v], outs[VARS[0]]))}" for v in VARS), flush=True)
        continue
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
