"""Per-kernel timing at the SDXL shapes of SURVEY.md §8(a) (run on the GPU box).

python tools/microbench.py [--dtype fp16|bf16] [--glds 0|1]
Prints one line per case: ms, TFLOP/s (or GB/s).  Random data, not zeros (cdna guide rule 25)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L


def timeit(fn, iters=20, warm=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--glds", type=int, default=1)
    ap.add_argument("--only", default="")
    ap.add_argument("--variant", type=int, default=0)
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda:0")
    L.lib().omg_debug_set_glds(a.glds)
    L.lib().omg_debug_set_gemm_variant(a.variant)
    R = lambda *s: torch.randn(*s, device=dev, dtype=dt)
    print(f"# dtype={a.dtype} glds={a.glds} variant={a.variant}")
    if not a.only or "gemm" in a.only:
        for (M, N, K, tag) in [(16384, 640, 640, "64^2 proj B4"), (16384, 1920, 640, "64^2 qkv B4"), (16384, 5120, 640, "64^2 geglu B4"),
                               (16384, 640, 2560, "64^2 ffout B4"), (4096, 1280, 1280, "32^2 proj B4"), (4096, 3840, 1280, "32^2 qkv B4"),
                               (4096, 10240, 1280, "32^2 geglu B4"), (4096, 1280, 5120, "32^2 ffout B4"), (2048, 1280, 1280, "32^2 proj B2"),
                               (2048, 10240, 1280, "32^2 geglu B2"), (2048, 1280, 5120, "32^2 ffout B2"), (8192, 1280, 1280, "32^2 proj B8"), (8192, 1280, 5120, "32^2 ffout B8"), (8192, 3840, 1280, "32^2 qkv B8"), (8192, 10240, 1280, "32^2 geglu B8"), (32768, 1280, 1280, "32^2 proj B32"), (8192, 8192, 8192, "square 8k"),
                               (4096, 4096, 4096, "square 4k")]:
            x, w = R(M, K), R(N, K)
            out = torch.empty(M, N, device=dev, dtype=dt)
            ms = timeit(lambda: ops.gemm(x, w, out=out))
            print(f"gemm {tag:16s} M={M:6d} N={N:6d} K={K:5d}: {ms:8.3f} ms  {2*M*N*K/ms/1e9:8.1f} TF/s")
    if not a.only or "conv" in a.only:
        for (B, H, C1, C2, Co, tag) in [(4, 128, 320, 0, 320, "128^2 320"), (4, 64, 640, 0, 640, "64^2 640"), (4, 32, 1280, 0, 1280, "32^2 1280"),
                                        (4, 32, 1280, 1280, 1280, "32^2 2560->1280"), (4, 128, 640, 320, 320, "128^2 960->320"), (2, 32, 1280, 0, 1280, "32^2 1280 B2")]:
            x1 = R(B, H, H, C1)
            x2 = R(B, H, H, C2) if C2 else None
            w = R(Co, 9 * (C1 + C2))
            ms = timeit(lambda: ops.conv2d(x1, w, 3, x2=x2))
            fl = 2 * B * H * H * Co * 9 * (C1 + C2)
            print(f"conv {tag:16s}: {ms:8.3f} ms  {fl/ms/1e9:8.1f} TF/s")
    if not a.only or "attn" in a.only:
        for (B, h, N, Nkv, tag) in [(4, 10, 4096, 4096, "self 64^2 B4"), (4, 20, 1024, 1024, "self 32^2 B4"), (2, 20, 1024, 1024, "self 32^2 B2"),
                                    (4, 10, 4096, 77, "cross 64^2 B4"), (4, 20, 1024, 77, "cross 32^2 B4")]:
            C = h * 64
            q, k, v = R(B, N, C), R(B, Nkv, C), R(B, Nkv, C)
            vt = ops.transpose_v(v, h)
            out = torch.empty(B, N, C, device=dev, dtype=dt)
            ms = timeit(lambda: ops.attention(q, k, vt, h, 0.125, out=out))
            fl = 4 * B * h * N * Nkv * 64
            ms_t = timeit(lambda: ops.transpose_v(v, h))
            print(f"attn {tag:16s}: {ms:8.3f} ms  {fl/ms/1e9:8.1f} TF/s   (transpose_v {ms_t:.3f} ms)")
    if not a.only or "norm" in a.only:
        for (B, HW, C, tag) in [(4, 16384, 320, "gn 128^2 320"), (4, 4096, 640, "gn 64^2 640"), (4, 1024, 1280, "gn 32^2 1280"), (4, 1024, 2560, "gn 32^2 2560")]:
            x = R(B, HW, C)
            g, b = R(C), R(C)
            ms = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, silu=True))
            print(f"{tag:21s}: {ms:8.3f} ms  {3*x.numel()*2/ms/1e6:8.1f} GB/s (2R+1W)")
        for (M, C) in [(16384, 640), (4096, 1280)]:
            x = R(M, C); g, b = R(C), R(C)
            ms = timeit(lambda: ops.layernorm(x, g, b, 1e-5))
            print(f"ln M={M} C={C}        : {ms:8.3f} ms  {2*x.numel()*2/ms/1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
