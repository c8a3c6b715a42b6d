"""Tile-variant sweep over the GEMM / conv shapes of the timed workload (run on the GPU box).

python tools/shape_sweep.py [--variants 1,12,14,11,13] [--batches 16,32]
For every shape prints the TF/s of each forced tile variant and the variant the heuristic picks; the table is the input
of `choose_variant` in omg_amd/csrc/gemm.hip."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L


def timeit(fn, iters=12, warm=2):
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
    ap.add_argument("--variants", default="1,12,14,11,13")
    ap.add_argument("--batches", default="16,32")
    ap.add_argument("--grouped", type=int, default=1, help="run plain GEMMs with one row group per sample (as the fused step does)")
    a = ap.parse_args()
    variants = [int(v) for v in a.variants.split(",")]
    dev, dt = torch.device("cuda:0"), torch.float16
    R = lambda *s: torch.randn(*s, device=dev, dtype=dt)
    lib = L.lib()
    print("# variants:", variants, " then: heuristic, torch.matmul, [N=1280: heuristic with bias+residual+weight slots]")
    for B in [int(b) for b in a.batches.split(",")]:
        gemms = [(4096, 640, 640, "64^2 proj"), (4096, 1920, 640, "64^2 qkv"), (4096, 5120, 640, "64^2 geglu-w"), (4096, 640, 2560, "64^2 ffout"),
                 (1024, 1280, 1280, "32^2 proj"), (1024, 3840, 1280, "32^2 qkv"), (1024, 10240, 1280, "32^2 geglu-w"), (1024, 1280, 5120, "32^2 ffout")]
        for hw, N, K, tag in gemms:
            M = B * hw
            x, w = R(M, K), R(N, K)
            out = torch.empty(M, N, device=dev, dtype=dt)
            row = []
            for v in variants + [0]:
                lib.omg_debug_set_gemm_variant(v)
                ms = timeit(lambda: ops.gemm(x, w, out=out, groups=B if a.grouped else 1))
                row.append(2 * M * N * K / ms / 1e9)
            # the same product through torch.matmul (hipBLASLt / rocBLAS): a library reference point, not a product path
            wt = w.t()
            ms = timeit(lambda: torch.matmul(x, wt, out=out))
            row.append(2 * M * N * K / ms / 1e9)
            if N == 1280:   # in-situ form of to_out: bias + residual + per-sample weight slots
                lib.omg_debug_set_gemm_variant(0)
                ws = torch.stack([w, w, w]).contiguous()
                ga = (torch.arange(B, device=dev) % 3).to(torch.int32)
                bias, res = R(N), R(M, N)
                ms = timeit(lambda: ops.gemm(x, ws, bias=bias, residual=res, out=out, groups=B, w_group_adapter=ga))
                row.append(2 * M * N * K / ms / 1e9)
            print(f"gemm B{B:<2d} {tag:14s} M={M:6d} N={N:5d} K={K:5d}: " + " ".join(f"{r:7.0f}" for r in row))
        convs = [(128, 320, 0, 320, 1, 0, "128^2 320"), (128, 320, 0, 320, 2, 0, "128^2 320 s2"), (64, 320, 0, 640, 1, 0, "64^2 320->640"),
                 (64, 640, 0, 640, 1, 0, "64^2 640"), (64, 640, 0, 640, 2, 0, "64^2 640 s2"), (32, 640, 0, 1280, 1, 0, "32^2 640->1280"),
                 (32, 1280, 0, 1280, 1, 0, "32^2 1280"), (32, 1280, 1280, 1280, 1, 0, "32^2 2560->1280"), (32, 1280, 640, 1280, 1, 0, "32^2 1920->1280"),
                 (32, 1280, 0, 1280, 1, 1, "32^2 1280 up"), (64, 1280, 640, 640, 1, 0, "64^2 1920->640"), (64, 640, 640, 640, 1, 0, "64^2 1280->640"),
                 (64, 640, 320, 640, 1, 0, "64^2 960->640"), (64, 640, 0, 640, 1, 1, "64^2 640 up"), (128, 640, 320, 320, 1, 0, "128^2 960->320"),
                 (128, 320, 320, 320, 1, 0, "128^2 640->320")]
        for H, C1, C2, Co, stride, up, tag in convs:
            x1 = R(B, H, H, C1)
            x2 = R(B, H, H, C2) if C2 else None
            w = R(Co, 9 * (C1 + C2))
            Ho = H * 2 if up else H // stride
            fl = 2 * B * Ho * Ho * Co * 9 * (C1 + C2)
            row = []
            for v in variants + [0]:
                lib.omg_debug_set_gemm_variant(v)
                ms = timeit(lambda: ops.conv2d(x1, w, 3, stride=stride, upsample=bool(up), x2=x2))
                row.append(fl / ms / 1e9)
            print(f"conv B{B:<2d} {tag:16s} M={B*Ho*Ho:6d} N={Co:5d} K={9*(C1+C2):5d}: " + " ".join(f"{r:7.0f}" for r in row))
    lib.omg_debug_set_gemm_variant(0)


if __name__ == "__main__":
    main()
