"""EXP builds (tools/exp/conv_out_v2.h): conv_out at the benchmark's batch (64 x 128 x 128 x 320 -> 4), product kernel | weights-in-registers kernel.
python tools/exp/conv_out_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from omg_amd import ops, _lib as L

lib, dev = L.lib(), torch.device("cuda:0")
if not hasattr(lib, "omg_debug_set_conv_out_variant"):
    raise SystemExit("needs an EXP build (make -C omg_amd/csrc EXP=1 DEV=1)")
for B in (64, 8):
    f = torch.randn(B, 128, 128, 320, device=dev, dtype=torch.float16)
    w = torch.randn(4, 3, 3, 320, device=dev, dtype=torch.float16) * 2880 ** -0.5
    b = torch.randn(4, device=dev, dtype=torch.float16)
    out = torch.empty(B, 4, 128, 128, device=dev, dtype=torch.float32)
    res = {}
    for var in (0, 2):
        lib.omg_debug_set_conv_out_variant(var)
        for _ in range(3):
            ops.conv_out(f, w, b, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            ops.conv_out(f, w, b, out=out)
        e.record(); torch.cuda.synchronize()
        res[var] = (s.elapsed_time(e) / 10, out.clone())
    lib.omg_debug_set_conv_out_variant(0)
    gb = f.numel() * 2 / 1e9
    print(f"conv_out {B}x128x128x320 -> 4: product {res[0][0] * 1e3:7.0f} us ({gb / res[0][0] * 1e3:5.2f} TB/s of input)   v2 {res[2][0] * 1e3:7.0f} us ({gb / res[2][0] * 1e3:5.2f} TB/s)   equal={torch.equal(res[0][1], res[2][1])}")
