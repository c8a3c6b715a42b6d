"""Static facts about the built kernels, read from the code objects inside libomg_hip.so (CPU, -m "not gpu"): what rocprofv3's
Scratch_Size / VGPR columns would show, checked on every build instead of in a profile.  Skipped when the library has not been built."""
import os
import re

import pytest

from tests import _codeobj

LIB = os.environ.get("OMG_CODEOBJ_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omg_amd", "csrc", "libomg_hip.so")
pytestmark = pytest.mark.skipif(not _codeobj.available(LIB), reason="libomg_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')")


@pytest.fixture(scope="module")
def ks():
    return _codeobj.kernels(LIB)


def pick(ks, *needles):
    return {n: k for n, k in ks.items() if all(s in n for s in needles)}


def test_every_kernel_is_wave64_without_dynamic_stack(ks):
    assert len(ks) > 100
    for n, k in ks.items():
        assert k["wavefront_size"] == 64, n
        assert not k.get("uses_dynamic_stack", False), n


def test_the_16_bit_gemm_instances_of_the_256_tile_use_no_scratch_at_all(ks):
    """One instance per epilogue form: a spill reload waits on vmcnt, i.e. on every store and LDS-DMA in flight,
    which cost the plain GEMM 20 % when the forms shared one kernel.  Round 4's gemm_kernel_v11 (table-driven K loop on five rotating
    half-stage buffers, 128 fragment VGPRs + 256 accumulators) keeps every form — bias-only (1), residual-through-LDS (2), GEGLU (3),
    gb / SiLU (4) — free of spills and scratch; the product build carries schedule 5 only."""
    seen = {}
    for n, k in pick(ks, "gemm_kernel_v11").items():
        m = re.search(r"Lb([01])ELi(\d)ELi(\d+)EEEv", n)
        conv, form, sch = int(m.group(1)), int(m.group(2)), int(m.group(3))
        seen.setdefault((form, sch), []).append(conv)
        assert k["agpr_count"] == 256 and k["max_flat_workgroup_size"] == 256, n        # 256 accumulators in AGPRs, one wave per SIMD
        if sch in (5, 10):                                                                  # the product schedule (10 = EXP builds: the same with the short prologue); EXP builds: schedule 8's GEGLU form parks two registers
            assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0 and k["sgpr_spill_count"] == 0, (n, k["private_segment_fixed_size"])
        assert k["private_segment_fixed_size"] <= 16, (n, k["private_segment_fixed_size"])
        assert k["group_segment_fixed_size"] == 0, n                                        # dynamic LDS only: 5 x 32 KB
    dev = not pick(ks, "gemm_kernel_v11", "IDF16b")                                     # make DEV=1: the f16 instances only
    assert all((f, 5) in seen and len(seen[(f, 5)]) == (2 if dev else 4) for f in (1, 2, 3, 4)), seen      # f16 / bf16 x Linear / conv
    for n, k in pick(ks, "gemm_kernel_v7", "Li0ELi2ELi5E").items():                # the 128 x 320 conv tile
        assert k["private_segment_fixed_size"] == 0, n


def test_attention_kernels_fit_two_workgroups_per_cu_without_scratch(ks):
    for name in ("attn_fwd_kernel3", "attn_fwd_kernel6", "attn_fwd_kernel2"):
        inst = pick(ks, name)
        assert len(inst) == 2, (name, list(inst))         # f16 and bf16
        for n, k in inst.items():
            assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, n
            assert k["vgpr_count"] + k.get("agpr_count", 0) <= 256, n      # 512 registers per SIMD lane: two waves
            assert k["group_segment_fixed_size"] <= 80 * 1024, n            # 160 KB of LDS: two workgroups


def test_scratch_users_are_known_and_small(ks):
    """Whatever spills must be on this list with a bound — a new entry is a regression to look at, not to wave through."""
    allowed = {"gemm_mx8_kernel": 128, "gemm_kernel_v7": 96, "gemm_kernel_v11": 16}       # bytes per lane.  v7 / v11: EXP builds only (round 3's XE forms and schedule 8: GEGLU's one or two registers); MX-fp8: its XE epilogue (4-26 VGPRs)
    for n, k in ks.items():
        sz = k["private_segment_fixed_size"]
        if sz:
            fam = [f for f in allowed if f in n]
            assert fam and sz <= allowed[fam[0]], (n, sz)


@pytest.mark.parametrize("family", ["gemm_kernel_v11", "gemm_kernel_v7", "gemm_mx8_kernel", "attn_fwd_kernel3", "attn_fwd_kernel6"])
def test_no_scratch_access_between_the_first_and_the_last_mfma(family):
    """Where the spilled registers of the table above are touched: never inside the MFMA region (K loop / key-tile loop).  A scratch
    reload there would wait on vmcnt and with it on the LDS-DMA of the next stage (any scratch use in a one-block-per-CU
    kernel costs far more than its traffic)."""
    dis = _codeobj.disassembly(LIB, family)
    assert dis
    for n, ins in dis.items():
        if "gemm_kernel_v7" in n and "Li4ELi4ELb1ELi0E" in n:
            continue                                       # form 0 of the XE tile: every form behind run-time tests, tools only
        mf = [i for i, x in enumerate(ins) if x.startswith("v_mfma")]
        assert mf, n
        inside = [ins[i] for i in range(mf[0], mf[-1]) if ins[i].startswith("scratch_")]
        assert not inside, (n, inside[:4])


def test_the_experimental_v12_kernels_put_their_loads_in_front_of_the_epilogue_stores(ks):
    """EXP builds only (tools/exp/gemm_v12.h, never in the product library): what the experiment is about must be true of the code hipcc
    emitted before anything is timed on it — no scratch in any form; in the prefetching forms (MODE & 4, EF != 2) the 32 LDS-DMA
    instructions of the next tile's first two stages, and in the early-residual forms (EF == 2) the 32 of the residual tile, sit between
    the barrier of the tile's last stage and the first store of its epilogue; the counted form (MODE 15, Linear) adds its 16 bias / group-bias loads there."""
    v12 = pick(ks, "gemm_kernel_v12")
    if not v12:
        pytest.skip("product build: no gemm_kernel_v12 (make -C omg_amd/csrc EXP=1)")
    for n, k in v12.items():
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (n, k["private_segment_fixed_size"])
        assert k["agpr_count"] == 256, n
    dis = _codeobj.disassembly(LIB, "gemm_kernel_v12")
    checked = 0
    for n, ins in dis.items():
        m = re.search(r"Lb([01])ELi(\d)ELi(\d+)EEEv", n)
        conv, form, mode = int(m.group(1)), int(m.group(2)), int(m.group(3))
        st = next(i for i, x in enumerate(ins) if x.startswith("buffer_store"))
        bar = max(i for i in range(st) if ins[i] == "s_barrier")
        loads = sum(1 for x in ins[bar:st] if x.startswith("buffer_load_dwordx4"))
        mfma = sum(1 for x in ins[bar:st] if x.startswith("v_mfma"))
        assert 32 <= mfma <= 40, (n, mfma)                    # the window: the 40 MFMAs behind the barrier (the first stores may overtake the last few)
        if form == 2:
            want = 32 if mode & 1 else 0
        elif mode & 4:
            want = 32 + (16 if (mode & 8) and not conv else 0)      # counted: + 8 bias and 8 group-bias loads (16 bytes per lane each)
        else:
            want = 0
        assert (loads >= want) if form == 4 else (loads == want), (n, loads, want)      # form 4 loads its per-row group bias inside the epilogue
        checked += 1
    assert checked == len(v12)


def test_the_experimental_256x320_tile_keeps_every_accumulator_where_the_source_pins_it(ks):
    """EXP builds only (tools/exp/gemm_v13.h, never in the product library).  The 256 x 320 tile has 320 accumulators per lane; with the MFMA
    builtin hipcc moved ~1000 of them between the AGPR and the VGPR half every stage and spilled (round 1 dropped the tile for that).  The
    experiment writes the MFMAs as inline asm with the allocation class in the constraint; what must then be true of the emitted code:
    no scratch, no spill; 240 MFMAs (three copies of the 80-MFMA stage), a fifth of them on VGPR accumulators; nothing but MFMAs, LDS reads,
    LDS-DMA and address arithmetic between the first and the last MFMA — no v_accvgpr_* at all; and, because the hazard recogniser does not
    see inline-asm MFMAs, the wait states of acc_fence directly behind the last MFMA and in front of the first one."""
    v13 = pick(ks, "gemm_kernel_v13")
    if not v13:
        pytest.skip("product build: no gemm_kernel_v13 (make -C omg_amd/csrc EXP=1)")
    for n, k in v13.items():
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0 and k["sgpr_spill_count"] == 0, (n, k["private_segment_fixed_size"])
        assert k["agpr_count"] == 256 and k["max_flat_workgroup_size"] == 256, n
        assert k["group_segment_fixed_size"] == 0, n                      # dynamic LDS only: 2 x 72 KB
    dis = _codeobj.disassembly(LIB, "gemm_kernel_v13", operands=True)
    assert len(dis) == len(v13)
    for n, ins in dis.items():
        mf = [i for i, x in enumerate(ins) if x.startswith("v_mfma")]
        assert len(mf) == 240, (n, len(mf))
        on_vgpr = sum(1 for i in mf if ins[i].split()[1].startswith("v["))
        assert on_vgpr == 48, (n, on_vgpr)                                # acc[i][4]: 4 of the 20 MFMAs of a k-step
        inside = ins[mf[0]:mf[-1]]
        bad = [x for x in inside if x.startswith(("v_accvgpr", "scratch_"))]
        assert not bad, (n, bad[:4])
        assert ins[mf[-1] + 1] == "s_nop 15" and ins[mf[-1] + 2] == "s_nop 15", (n, ins[mf[-1] + 1:mf[-1] + 4])
        last_write = max(i for i in range(mf[0]) if ins[i].startswith("v_accvgpr_write"))      # the initialisation of the AGPR accumulators
        assert ins[last_write:mf[0]].count("s_nop 15") >= 2, n           # ... is in front of the first fence


def test_the_experimental_row_major_v_attention_fits_two_workgroups_per_cu(ks):
    """EXP builds only (tools/exp/attn_v7.h): attn_fwd_kernel7 = v3 with the V tile staged row-major and transposed on the way out of LDS.
    It must keep v3's occupancy (two workgroups per CU: <= 256 registers, <= 80 KB of LDS, no scratch) and read every V^T fragment with two
    ds_read_b64_tr_b16 — 16 per key tile, in each of the three copies of the tile body (first tile, steady state, ragged tail) — where v3 has
    eight ds_read_b128; the K fragments stay on ds_read_b128."""
    v7 = pick(ks, "attn_fwd_kernel7")
    if not v7:
        pytest.skip("product build: no attn_fwd_kernel7 (make -C omg_amd/csrc EXP=1)")
    assert len(v7) == 6                                  # f16 / bf16 x (7 | 8 = three-address asm first MFMA, Q loads up front | 9 = 8 + tools-only knobs)
    for n, k in v7.items():
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, n
        assert k["vgpr_count"] + k.get("agpr_count", 0) <= 256 and k["group_segment_fixed_size"] <= 80 * 1024, n
    for n, ins in _codeobj.disassembly(LIB, "attn_fwd_kernel7").items():
        assert ins.count("ds_read_b64_tr_b16") == 48 and ins.count("ds_read_b128") == 24, (n, ins.count("ds_read_b64_tr_b16"), ins.count("ds_read_b128"))
        if "Lb1ELb" in n:                                # the asm form: no 16-register copies of the reference-maximum splat in front of the tile's MFMAs
            assert ins.count("v_mov_b64_e32") <= 64 and ins.count("v_mov_b32_e32") <= 200, (n, ins.count("v_mov_b64_e32"), ins.count("v_mov_b32_e32"))


def test_the_short_prologue_reaches_its_first_lds_dma_behind_fewer_round_trips(ks):
    """EXP builds only (gemm_v11.h, SCH == 10 = variant 31).  The product kernel of the 256 x 256 tile fetches its launch parameters field by field —
    five or six `s_waitcnt` in front of the first LDS-DMA, one of them on the VECTOR-memory load of the group's adapter id — once per tile.  The experiment
    requests them in one batch and reads the adapter id through the scalar cache: at most four waits (Linear: three), no vector load in front of the
    first DMA.  Product builds: those waits are asserted instead, so that the finding stays true of what is shipped until the experiment lands."""
    dis = _codeobj.disassembly(LIB, "gemm_kernel_v11", operands=True)

    def way_to_first_dma(ins):
        first = next(i for i, x in enumerate(ins) if x.startswith("buffer_load_dwordx4") and x.endswith("lds"))
        head = ins[:first]
        return sum(1 for x in head if x.startswith("s_waitcnt")), sum(1 for x in head if x.startswith(("global_load", "flat_load")))

    short = {n: way_to_first_dma(i) for n, i in dis.items() if "ELi10EEEv" in n}
    prod = {n: way_to_first_dma(i) for n, i in dis.items() if "ELi5EEEv" in n}
    assert prod and all(w >= 5 and v == 1 for w, v in prod.values()), prod
    if not short:
        pytest.skip("product build: no SCH == 10 instance of gemm_kernel_v11 (make -C omg_amd/csrc EXP=1)")
    assert all(w <= 4 and v == 0 for w, v in short.values()), short
