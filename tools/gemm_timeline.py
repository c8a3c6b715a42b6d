"""Per-CU block timeline of the v7 GEMM kernel (debug bit 16): prologue / main loop / epilogue-issue durations and the
gap between consecutive blocks on the same CU.  python tools/gemm_timeline.py [M N K] [extra dbg bits]"""
import ctypes, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops, _lib as L

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 10240, 1280)
extra = int(sys.argv[4]) if len(sys.argv) > 4 else 0
lib = L.lib()
lib.omg_debug_read_ts.restype = ctypes.c_int
lib.omg_debug_read_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev, dtype=torch.float16)
w = torch.randn(N, K, device=dev, dtype=torch.float16)
out = torch.empty(M, N, device=dev, dtype=torch.float16)
RES = torch.randn(M, N, device=dev, dtype=torch.float16) if len(sys.argv) > 6 and sys.argv[6] == "res" else None
var = int(sys.argv[5]) if len(sys.argv) > 5 else 15
lib.omg_debug_set_gemm_variant(var | ((16 | extra) << 8))
for _ in range(3):
    ops.gemm(x, w, out=out, residual=RES)
torch.cuda.synchronize()
nb = min(8192, ((M + 255) // 256) * ((N + 255) // 256))
buf = (ctypes.c_longlong * (6 * nb))()
assert lib.omg_debug_read_ts(buf, nb) == 0
rows = [[buf[i * 6 + j] for j in range(6)] for i in range(nb)]
t_min = min(r[0] for r in rows)
cus = defaultdict(list)
for b, r in enumerate(rows):
    cus[(r[5] & 0xf, r[4] & 0xff00)].append((r[0] - t_min, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[3] - t_min, b))
print(f"{nb} blocks on {len(cus)} (xcc, se/sh/cu/simd-pipe) slots; ticks of 10 ns")
pro, main, epi, gap = [], [], [], []
for k, v in cus.items():
    v.sort()
    for i, e in enumerate(v):
        pro.append(e[1]); main.append(e[2]); epi.append(e[3])
        if i:
            gap.append(e[0] - v[i - 1][4])
mean = lambda a: sum(a) / max(len(a), 1)
print(f"prologue (start -> stage 0 landed) {mean(pro)/100:.2f} us   main loop {mean(main)/100:.2f} us   epilogue issue {mean(epi)/100:.2f} us   "
      f"gap to next block on the slot {mean(gap)/100:.2f} us   kernel span {(max(r[3] for r in rows) - t_min)/100:.1f} us")
k0 = sorted(cus)[0]
print("one slot:", [(e[0], e[1], e[2], e[3]) for e in cus[k0][:6]])
lib.omg_debug_set_gemm_variant(var | (extra << 8))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.gemm(x, w, out=out, residual=RES)
s.record()
for _ in range(10):
    ops.gemm(x, w, out=out, residual=RES)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"untimed-stamp launch: {ms*1000:.1f} us = {2*M*N*K/ms/1e9:.0f} TF/s")
lib.omg_debug_set_gemm_variant(0)
