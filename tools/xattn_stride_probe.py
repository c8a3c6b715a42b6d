"""Cross-attention kernel (attn_fwd_kernel6, <= 128 keys): is it bound by the 128-byte-per-(row, head) access pattern of Q and O?
The same bytes and FLOPs three ways: (64, 20, 1024, 77) as the UNet has it (a head = a 128-byte column slice of 2560-byte rows), the same with heads
= 1 and 1280 samples (every row one contiguous 128-byte line, rows back to back), and 16 keys.   python tools/xattn_stride_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omg_amd import ops

dev, dt = torch.device("cuda:0"), torch.float16


def timeit(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for (B, heads, Nq, Nkv) in [(64, 20, 1024, 77), (1280, 1, 1024, 77), (64, 20, 1024, 16), (1280, 1, 1024, 16), (64, 10, 4096, 77), (640, 1, 4096, 77)]:
    C = heads * 64
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(B, Nq, C, device=dev, generator=g).to(dt)
    k = torch.randn(B, Nkv, C, device=dev, generator=g).to(dt)
    v = torch.randn(B, Nkv, C, device=dev, generator=g).to(dt)
    vt = ops.transpose_v(v, heads)
    o = torch.empty_like(q)
    # a second buffer set so that consecutive iterations do not find Q / O in the Infinity Cache
    q2, o2 = q.clone(), torch.empty_like(q)
    state = [0]

    def run():
        state[0] ^= 1
        ops.attention(q2 if state[0] else q, k, vt, heads, 0.125, out=o2 if state[0] else o)

    us = timeit(run)
    byt = 2 * q.numel() * 2
    # the self-attention kernel (attn_fwd_kernel7: 64 query rows per wave, K / V tiles by LDS-DMA, row-major V) forced onto the same problem
    from omg_amd import _lib as L
    ref = o.clone() if not state[0] else o2.clone()
    L.lib().omg_debug_set_attn_variant(7)
    vr = ops.RowMajorV.__new__(ops.RowMajorV)      # the class refuses <= 128 keys: this probe is the experiment that decides whether it should
    vr.v = v
    o7 = torch.empty_like(q)

    def run7():
        state[0] ^= 1
        ops.attention(q2 if state[0] else q, k, vr, heads, 0.125, out=o7)

    us7 = timeit(run7)
    L.lib().omg_debug_set_attn_variant(0)
    print(f"    attn_fwd_kernel7 on it: {us7:7.1f} us   max |v7 - v6| {(o7.float() - ref.float()).abs().max().item():.2e}")
    print(f"({B},{heads},{Nq},{Nkv}): {us:7.1f} us   Q + O {byt / 1e6:.0f} MB -> {byt / us / 1e6:.2f} TB/s   {4.0 * B * heads * Nq * Nkv * 64 / us / 1e6:.0f} TF/s")
