"""Clock / power of the GPU while the hot kernels run back to back (rocm-smi sampled from a side thread): is the MFMA rate the
in-situ GEMM sees bounded by the power cap?   python tools/power_probe.py"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L

samples, stop, phase = [], False, ["idle"]
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "-c", "-P", "-t", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(o)
            c = d.get("card0", {})
            samples.append((time.time(), phase[0], {k: v for k, v in c.items() if any(s in k.lower() for s in ("sclk", "power", "junction", "mclk"))}))
        except Exception as e:      # noqa
            samples.append((time.time(), phase[0], {"error": str(e)[:80]}))
        time.sleep(0.25)
th = threading.Thread(target=sampler); th.start()
dev = torch.device("cuda:0")
def burn(name, fn, flops, secs=4.0):
    fn(); torch.cuda.synchronize()
    phase[0] = name
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(10): fn()
        torch.cuda.synchronize(); n += 10
    dt = time.time() - t0
    phase[0] = "idle"
    print(f"{name}: {flops * n / dt / 1e12:.0f} TF/s sustained over {dt:.1f} s")
    time.sleep(1.0)
time.sleep(1.0)
M, N, K = 65536, 10240, 1280
x = torch.randn(M, K, device=dev, dtype=torch.float16); w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
out = torch.empty(M, N // 2, device=dev, dtype=torch.float16)
burn("gemm_geglu_random", lambda: ops.gemm(x, w, act=L.ACT_GEGLU, out=out), 2 * M * N * K)
x0 = torch.zeros_like(x)
burn("gemm_geglu_zero_A", lambda: ops.gemm(x0, w, act=L.ACT_GEGLU, out=out), 2 * M * N * K)
xs = (torch.randn(M, K, device=dev) * 0.02).to(torch.float16)
burn("gemm_geglu_small_A", lambda: ops.gemm(xs, w, act=L.ACT_GEGLU, out=out), 2 * M * N * K)
xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
outb = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
burn("gemm_geglu_bf16_random", lambda: ops.gemm(xb, wb, act=L.ACT_GEGLU, out=outb), 2 * M * N * K)
outf = torch.empty(M, N, device=dev, dtype=torch.float16)
burn("torch_matmul_bench_shape", lambda: torch.matmul(x, w.t(), out=outf), 2 * M * N * K)
del outf
xq, wq = ops.quant_mx8(x), ops.quant_mx8(w)
burn("gemm_mx8_geglu_random", lambda: ops.gemm_mx8(xq, wq, act=L.ACT_GEGLU, out=out), 2 * M * N * K)
a = torch.randn(8192, 8192, device=dev, dtype=torch.float16); b = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
burn("torch_matmul_8192", lambda: torch.matmul(a, b), 2 * 8192 ** 3)
q = torch.randn(64, 4096, 640, device=dev, dtype=torch.float16)
vt = ops.transpose_v(q, 10)
burn("attn_64x10x4096", lambda: ops.attention(q, q, vt, 10, 0.125), 4 * 64 * 10 * 4096 * 4096 * 64)
stop = True; th.join()
by = {}
for t, ph, d in samples:
    by.setdefault(ph, []).append(d)
for ph, ds in by.items():
    keys = sorted({k for d in ds for k in d})
    print(ph, len(ds), "samples")
    for k in keys:
        vals = [d[k] for d in ds if k in d]
        print("   ", k, vals[:3], "...", vals[-3:])
