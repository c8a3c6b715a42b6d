"""Per-phase cycle totals of block 0's waves in the staggered GEMM loop (dbg bit 16).  python tools/gemm_phases.py M N K"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L
M, N, K = (int(a) for a in sys.argv[1:4])
lib = L.lib()
lib.omg_debug_read_cycles.restype = ctypes.c_int
lib.omg_debug_read_cycles.argtypes = [ctypes.c_void_p]
lib.omg_debug_set_gemm_variant(9 + 256 * 16)
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev, dtype=torch.float16); w = torch.randn(N, K, device=dev, dtype=torch.float16)
out = torch.empty(M, N, device=dev, dtype=torch.float16)
for _ in range(3):
    ops.gemm(x, w, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
print("rc", lib.omg_debug_read_cycles(buf))
print("wave  wait_vmcnt  barrier     mma     dma_issue  ds_read(+wait)   per-stage: wait bar mma dma rd   (nk)")
for w_ in range(8):
    r = [buf[w_ * 8 + i] for i in range(6)]
    nk = max(r[5], 1)
    print(w_, r[:5], " | ", " ".join(f"{v / nk:7.0f}" for v in r[:5]), f" ({nk})")
