"""The hand-written GEMM against the vendor library on the benchmark's own Linear shapes, same box, same data, interleaved
(MI355X; library = torch.matmul -> hipBLASLt; a reference point for tools/ only, never a product path).

python tools/vs_hipblaslt.py [--rounds 5]        prints per shape: ours (plain epilogue), hipBLASLt, ratio; ours in situ form (bias / residual / GEGLU)
python tools/vs_hipblaslt.py --names             one matmul per shape, for `rocprofv3 --kernel-trace --stats` to name the library's kernels
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L

SHAPES = [(65536, 10240, 1280, "geglu"), (65536, 1280, 5120, "res"), (65536, 1280, 1280, "res"), (65536, 3840, 1280, ""),
          (32768, 10240, 1280, "geglu"), (262144, 5120, 640, "geglu"), (262144, 640, 640, "res"), (262144, 640, 2560, "res"),
          (262144, 1920, 640, ""), (32768, 1280, 1280, "res")]


def t(fn, n=8):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--names", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for M, N, K, kind in SHAPES:
        x = torch.randn(M, K, device=dev, dtype=torch.float16)
        w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
        wt = w.t()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        if a.names:
            torch.matmul(x, wt, out=out); torch.cuda.synchronize()
            continue
        b = torch.randn(N, device=dev, dtype=torch.float16)
        res = torch.randn(M, N, device=dev, dtype=torch.float16) if kind == "res" else None
        out2 = torch.empty(M, N // 2 if kind == "geglu" else N, device=dev, dtype=torch.float16)
        ours, lib_, situ = [], [], []
        for _ in range(a.rounds):
            ours.append(t(lambda: ops.gemm(x, w, out=out)))
            lib_.append(t(lambda: torch.matmul(x, wt, out=out)))
            situ.append(t(lambda: ops.gemm(x, w, bias=b, residual=res, act=L.ACT_GEGLU if kind == "geglu" else 0, out=out2)))
        f = 2.0 * M * N * K / 1e9
        med = lambda v: sorted(v)[len(v) // 2]
        print(f"{M:7d} x {N:5d} x {K:4d}  ours {f / med(ours):7.0f} TF/s   hipBLASLt {f / med(lib_):7.0f} TF/s   ours/lib {med(lib_) / med(ours):5.2f}   "
              f"in situ ({kind or 'bias':5s}) {f / med(situ):7.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
