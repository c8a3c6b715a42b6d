"""Generates omg_amd/csrc/gemm_v12_sched.inc: straight-line K-loop stage bodies of the 256x256 four-wave GEMM
(gemm_v12.h) from schedule tables.  A stage is 32 slots of two MFMAs; a table places the 32 fragment reads, the 16 LDS-DMA
instructions and the barrier.  The counted `s_waitcnt lgkmcnt(N)` in front of each k-step is derived here (LDS returns in order).
    python tools/gen_ksched.py            # rewrites the .inc (committed; hipcc needs no python at build time)
Rules checked: a read of set k in a slot < 8k serves THIS stage (current buffer, before the barrier); in a slot >= 8k+8 the NEXT stage
(other buffer, behind the barrier); DMAs refill the current buffer behind the barrier."""
import os

def sched(i):
    rd = {}
    for r in range(8):
        if i in (0, 8): # round 3's hand-written schedule: every set two k-steps ahead, barrier between k-steps 1 and 2
            rd[(2, r)] = r; rd[(3, r)] = 8 + r; rd[(0, r)] = 16 + r; rd[(1, r)] = 24 + r
        else:           # early release: both remaining sets of the current buffer are read in k-step 0 (two reads per slot)
            rd[(2, r)] = r; rd[(3, r)] = r; rd[(0, r)] = 16 + r; rd[(1, r)] = 24 + r
    bar = {0: 16, 1: 12, 2: 16, 3: 12, 4: 14, 5: 12, 6: 12, 7: 12, 8: 16, 9: 10}[i]
    dma = [bar + d // 2 if i == 3 else bar + d for d in range(16)]
    # 5..7: FIVE 32 KB half-stage buffers (A cur, W cur, A next, W next, spare) instead of two 64 KB stages: the A half of stage kt + 2 goes
    # into the spare from slot 0 on, its W half into the current W buffer behind the barrier, so the 16 DMAs can be spread over the WHOLE
    # stage (one per two slots = one per four MFMAs) instead of one per slot behind the barrier
    if i in (5, 8, 9): dma = [2 * d for d in range(16)]       # 8: the ring with one read per slot all stage; 9: barrier two slots earlier
    if i == 6: dma = [2 * d + 1 for d in range(16)]
    if i == 7: dma = [bar + d for d in range(16)]            # control: the ring with the DMAs bunched as in schedule 1
    return rd, bar, dma

RING = (5, 6, 7, 8, 9)

def check(rd, bar, dma, ring):
    for (k, r), s in rd.items():
        assert 0 <= s < 32
        if s < 8 * k: assert s < bar, (k, r, s)
        else: assert s >= 8 * k + 8 and s >= bar, (k, r, s)
    # two stages: every DMA refills the current buffer (behind the barrier); ring: only the W half does, the A half goes into the spare
    assert all((0 if ring and d < 8 else bar) <= sl < 32 for d, sl in enumerate(dma)) and 1 <= bar < 32
    assert dma == sorted(dma)                                  # vmcnt counts in issue order

def order(rd):       # program order of the reads of one stage
    return sorted(rd, key=lambda kr: (rd[kr], kr[0], kr[1]))

def wait_count(rd, k, has1):
    o = order(rd)
    last = max(o.index((k, r)) for r in range(8))
    cur = rd[o[last]] < 8 * k
    exists = lambda kr: rd[kr] < 8 * kr[0] or has1        # reads serving the next stage are skipped when there is none
    if cur:
        n = sum(1 for kr in o[last + 1:] if rd[kr] < 8 * k and exists(kr))
    else:               # set k was read in the previous stage (which had a successor: all its reads exist)
        n = len(o) - 1 - last + sum(1 for kr in o if rd[kr] < 8 * k and exists(kr))
    return min(n, 14)      # 15 is the field maximum = "no wait": never rely on a saturated counter

def emit(i):
    rd, bar, dma = sched(i)
    ring = i in RING
    check(rd, bar, dma, ring)
    o = order(rd)
    early = sum(1 for sl in dma if sl < bar)                   # DMAs of stage kt + 2 already in flight at the barrier
    L = [f"// schedule {i}: barrier in front of slot {bar}; reads per slot " + " ".join(str(sum(1 for v in rd.values() if v == s)) for s in range(32))
         + "; DMAs per slot " + " ".join(str(dma.count(s)) for s in range(32))]
    L.append(f"#define OMG_KS_PROLOGUE_{i}() do {{ \\")
    for kr in o:
        if rd[kr] >= 8 * kr[0] + 8: L.append(f"  OMG_XRD1({kr[0]}, {kr[1]}, 0u); \\")
    L.append("} while (0)")
    L.append(f"#define OMG_KS_STAGE_{i}(HAS1_, HAS2_) do {{ \\")
    for s in range(32):
        k, q = divmod(s, 8)
        if s == 0 and early: L.append("  if (HAS2_) OMG_PREP(kt + 2); \\")
        if s == bar:
            if early:
                L.append(f'  if (HAS2_) asm volatile("s_waitcnt vmcnt({early}) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); \\')
            else:
                L.append('  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); if (HAS2_) OMG_PREP(kt + 2); \\')
        if q == 0:
            L.append(f"  if (HAS1_) OMG_XWAIT({k}, {wait_count(rd, k, True)}); else OMG_XWAIT({k}, {wait_count(rd, k, False)}); \\")
        L.append(f"  OMG_XMM1({k}, {2 * q}); OMG_SB; \\")
        for kr in o:
            if rd[kr] == s:
                if s < 8 * kr[0]: L.append(f"  OMG_XRD1({kr[0]}, {kr[1]}, tcur); \\")
                else: L.append(f"  if (HAS1_) OMG_XRD1({kr[0]}, {kr[1]}, tnxt); \\")
        L.append(f"  OMG_SB; OMG_XMM1({k}, {2 * q + 1}); OMG_SB; \\")
        for d in range(16):
            if dma[d] == s: L.append(f"  if (HAS2_) OMG_DMA({d}, curb); \\" if not ring else f"  if (HAS2_) OMG_DMAR({d}); \\")
        L.append("  OMG_SB; \\")
    L.append("} while (0)")
    return "\n".join(L)

def emit_last(i):
    """The LAST stage of a tile (no successor: no fragment reads for a next stage, no DMAs of this tile) with HOOKS for gemm_v12.h: OMG_HEAD(n)
    behind MFMA n in front of the barrier, OMG_TAIL(n) behind MFMA n after it.  The MFMAs are OMG_XMML (the kernel anchors each one where it
    is written: with no fragment read of a next stage behind them nothing else keeps them from sinking to the end of the block).  From the barrier on every LDS buffer of the tile is free (all
    fragment reads of the stage are issued in front of it and waited for by its lgkmcnt(0)): the tail is where the residual tile of THIS tile's
    epilogue or the first two stages of the NEXT tile can be put in flight under 40 MFMAs."""
    rd, bar, dma = sched(i)
    assert i in RING
    o = order(rd)
    L = [f"// schedule {i}, last stage of a tile: {2 * bar} head hooks, {64 - 2 * bar} tail hooks"]
    L.append(f"#define OMG_KS_LAST_{i}() do {{ \\")
    n = 0
    for s in range(32):
        k, q = divmod(s, 8)
        if s == bar:
            L.append('  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); \\')
            n = 0
        if q == 0:
            L.append(f"  OMG_XWAIT({k}, {wait_count(rd, k, False)}); \\")
        hook = "OMG_HEAD" if s < bar else "OMG_TAIL"
        L.append(f"  OMG_XMML({k}, {2 * q}); OMG_SB; \\")
        for kr in o:
            if rd[kr] == s and s < 8 * kr[0]: L.append(f"  OMG_XRD1({kr[0]}, {kr[1]}, tcur); \\")
        L.append(f"  {hook}({n}); OMG_SB; \\"); n += 1
        L.append(f"  OMG_XMML({k}, {2 * q + 1}); OMG_SB; \\")
        L.append(f"  {hook}({n}); OMG_SB; \\"); n += 1
    L.append("} while (0)")
    return "\n".join(L), 2 * bar, 64 - 2 * bar


if __name__ == "__main__":
    # Round 5: the ring (schedule 5) won round 4's A/B and gemm_kernel_v12 (its persistent form) won round 5's: the product carries schedule 5 only.
    # `python tools/gen_ksched.py 0 1 9` still emits other tables (to stdout) for anyone who wants to time one again.
    import sys
    if len(sys.argv) > 1:
        print("\n".join(emit(int(a)) for a in sys.argv[1:]))
        raise SystemExit(0)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omg_amd", "csrc", "gemm_v12_sched.inc")
    with open(out, "w") as f:
        body, nh, nt = emit_last(5)
        f.write("// GENERATED by tools/gen_ksched.py — do not edit.  Stage bodies of gemm_kernel_v12 (gemm_v12.h): prologue reads, steady-state stage, last stage.\n")
        f.write(emit(5) + "\n")
        f.write(f"#define OMG_KS_LAST_HEADS {nh}\n#define OMG_KS_LAST_TAILS {nt}\n")
        f.write(body + "\n")
    print("wrote", out)
