"""MX-fp8 GEMM vs the fp16 GEMM on the Linear shapes of the benchmark (run on the GPU box).  python tools/mx8_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L


def timeit(fn, iters=10, warm=2):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev, dt = torch.device("cuda:0"), torch.float16
    lib = L.lib()
    shapes = [(1024, 10240, 1280, "32^2 geglu"), (1024, 1280, 5120, "32^2 ffout"), (1024, 3840, 1280, "32^2 qkv"), (1024, 1280, 1280, "32^2 proj"),
              (4096, 5120, 640, "64^2 geglu"), (4096, 640, 2560, "64^2 ffout"), (4096, 1920, 640, "64^2 qkv"), (4096, 640, 640, "64^2 proj")]
    print("# shape: fp16 TF/s | mx8 TF/s: one tile per block at DMA split 8 / 12 / 16, 12 with the register-direct epilogue, and (round 6) the PERSISTENT "
          "tile walk (gemm_mx8_kernel_p; the launcher picks it for N <= 1280, K <= 1280) | quantiser us (GB/s)")
    for hw, N, K, tag in shapes + [(8192 // B if B <= 8192 else 1, 8192, 8192, "8192^3")]:
        M = B * hw
        x = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt)
        geglu = "geglu" in tag
        act = L.ACT_GEGLU if geglu else L.ACT_NONE
        out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt)
        fl = 2.0 * M * N * K
        t16 = timeit(lambda: ops.gemm(x, w, out=out, act=act))
        xq, wq = ops.quant_mx8(x), ops.quant_mx8(w)
        row = []
        for d1 in (8 | (128 << 8), 12 | (128 << 8), 16, 12 | (64 << 8), 12 | (512 << 8)):      # dbg bit 128 = one tile per block, 512 = the persistent walk, whatever the shape
            lib.omg_debug_set_mx8_split(d1)
            row.append(fl / timeit(lambda: ops.gemm_mx8(xq, wq, out=out, act=act)) / 1e9)
        lib.omg_debug_set_mx8_split(12)
        tq = timeit(lambda: ops.quant_mx8(x, out=xq))
        print(f"{tag:12s} M={M:7d} N={N:5d} K={K:5d}: {fl / t16 / 1e9:7.0f} | " + " ".join(f"{r:7.0f}" for r in row) + f" | {tq * 1e3:7.1f} us ({M * K * 3 / tq / 1e6:6.0f} GB/s)")


if __name__ == "__main__":
    main()
