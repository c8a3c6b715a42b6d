"""Is the 15 % the residual-form GEMMs lose IN SITU (65536 x 1280 x 1280 + residual 1139 TF/s alone, 964 in the benchmark step) the
kernel's, or the memory system's?  Alone, a microbenchmark re-reads the same A / residual / C buffers every launch — 0.5 GB that mostly lives in the
256 MB Infinity Cache; in the step the A operand has just been written by another kernel, the residual was last touched several kernels ago, and both
come from HBM.  This tool times the same launch HOT (one buffer set) and COLD (a ring of buffer sets larger than the cache, so every launch reads
operands nobody has touched for > 4 GB of traffic), ours and the vendor library's plain GEMM, interleaved on one box.
    python tools/gemm_cold.py [--rounds 3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops

SHAPES = [(65536, 1280, 1280, True), (65536, 1280, 5120, True), (65536, 3840, 1280, False), (262144, 640, 640, True), (32768, 1280, 1280, True)]


def timed(fns, n):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fns[i % len(fns)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print("# M x N x K (+res): ours hot | ours cold | hipBLASLt plain hot | hipBLASLt plain cold   [TF/s];  arithmetic intensity FLOP per byte of A + residual + C")
    for M, N, K, res in SHAPES:
        per_set = (M * K + M * N * (2 if res else 1)) * 2
        nset = max(2, int(5e9 // per_set))                       # >= 5 GB of distinct operands: 20 x the Infinity Cache
        w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
        b = torch.randn(N, device=dev, dtype=torch.float16)
        sets = [(torch.randn(M, K, device=dev, dtype=torch.float16), torch.randn(M, N, device=dev, dtype=torch.float16) if res else None,
                 torch.empty(M, N, device=dev, dtype=torch.float16)) for _ in range(nset)]
        ours = [lambda s_=s_: ops.gemm(s_[0], w, bias=b, residual=s_[1], out=s_[2]) for s_ in sets]
        lib_ = [lambda s_=s_: torch.matmul(s_[0], w.t(), out=s_[2]) for s_ in sets]
        f = 2.0 * M * N * K / 1e9
        med = lambda v: sorted(v)[len(v) // 2]
        r = {k: [] for k in ("oh", "oc", "lh", "lc")}
        for _ in range(a.rounds):
            r["oh"].append(timed(ours[:1], 2 * nset)); r["oc"].append(timed(ours, 2 * nset))
            r["lh"].append(timed(lib_[:1], 2 * nset)); r["lc"].append(timed(lib_, 2 * nset))
        print(f"{M:7d} x {N:5d} x {K:4d} {'+res' if res else '    '}  {f / med(r['oh']):6.0f} | {f / med(r['oc']):6.0f} | {f / med(r['lh']):6.0f} | {f / med(r['lc']):6.0f}"
              f"    {2.0 * M * N * K / per_set:5.0f} FLOP/B ({nset} buffer sets of {per_set / 1e6:.0f} MB)", flush=True)
        del sets, ours, lib_


if __name__ == "__main__":
    main()
