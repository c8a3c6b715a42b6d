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
    which cost the plain GEMM 20 % when the forms shared one kernel.  gemm_kernel_v12 (table-driven K loop on five rotating half-stage
    buffers, 128 fragment VGPRs + 256 accumulators, persistent tile walk) keeps every form — bias-only (1), residual-through-LDS (2),
    GEGLU (3), gb / SiLU (4) — free of spills and scratch."""
    seen = {}
    for n, k in pick(ks, "gemm_kernel_v12").items():
        m = re.search(r"Lb([01])ELi(\d)EEEv", n)
        conv, form = int(m.group(1)), int(m.group(2))
        seen.setdefault(form, []).append(conv)
        assert k["agpr_count"] == 256 and k["max_flat_workgroup_size"] == 256, n        # 256 accumulators in AGPRs, one wave per SIMD
        # no scratch: the tile walk's scalar state may be parked in VGPR LANES (v_writelane / v_readlane, `sgpr_spill_count`), never in memory
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (n, k["private_segment_fixed_size"])
        assert k["group_segment_fixed_size"] == 0, n                                        # dynamic LDS only: 5 x 32 KB
    dev = not pick(ks, "gemm_kernel_v12", "IDF16b")                                     # make DEV=1: the f16 instances only
    assert all(f in seen and len(seen[f]) == (2 if dev else 4) for f in (1, 2, 3, 4)), seen      # f16 / bf16 x Linear / conv
    for n, k in pick(ks, "gemm_kernel_v7", "Li0ELi2ELi5E").items():                # the 128 x 320 conv tile
        assert k["private_segment_fixed_size"] == 0, n


def test_attention_kernels_fit_two_workgroups_per_cu_without_scratch(ks):
    for name in ("attn_fwd_kernel7", "attn_fwd_kernel6", "attn_fwd_kernel2"):
        inst = pick(ks, name)
        assert len(inst) == 2, (name, list(inst))         # f16 and bf16
        for n, k in inst.items():
            assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, n
            assert k["vgpr_count"] + k.get("agpr_count", 0) <= 256, n      # 512 registers per SIMD lane: two waves
            assert k["group_segment_fixed_size"] <= 80 * 1024, n            # 160 KB of LDS: two workgroups


def test_scratch_users_are_known_and_small(ks):
    """Whatever spills must be on this list with a bound — a new entry is a regression to look at, not to wave through."""
    allowed = {"gemm_mx8_kernel": 128}       # bytes per lane: the MX-fp8 kernel's XE epilogue (4-26 VGPRs)
    for n, k in ks.items():
        sz = k["private_segment_fixed_size"]
        if sz:
            fam = [f for f in allowed if f in n]
            assert fam and sz <= allowed[fam[0]], (n, sz)


@pytest.mark.parametrize("family", ["gemm_kernel_v12", "gemm_kernel_v13", "gemm_kernel_v7", "gemm_mx8_kernel", "attn_fwd_kernel7", "attn_fwd_kernel6"])
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
        if "gemm_mx8_kernel_p" in n:
            # the persistent MX-fp8 kernel (round 6): the tile hand-over — next tile's coordinates, descriptors, accumulator initialisation — sits BETWEEN the
            # stage bodies in address order and may touch its few spilled registers; what must stay clean is every stage body itself (a run of MFMAs
            # less than 40 instructions apart)
            runs, start = [], mf[0]
            for a_, b_ in zip(mf, mf[1:] + [None]):
                if b_ is None or b_ - a_ > 40:
                    runs.append((start, a_))
                    start = b_
            assert len(runs) >= 3 and all(sum(1 for i in mf if r0 <= i <= r1) >= 16 for r0, r1 in runs), (n, runs)
            inside = [ins[i] for r0, r1 in runs for i in range(r0, r1) if ins[i].startswith("scratch_")]
        else:
            inside = [ins[i] for i in range(mf[0], mf[-1]) if ins[i].startswith("scratch_")]
        assert not inside, (n, inside[:4])


def test_the_256_tile_puts_its_loads_in_front_of_the_epilogue_stores(ks):
    """gemm_kernel_v12 (csrc/gemm_v12.h): what the tile walk is about must be true of the code hipcc emitted — in the persistent forms
    (EF != 2) the 32 LDS-DMA instructions of the next tile's first two stages, and in the residual form (EF == 2) the 32 of the residual
    tile, sit between the barrier of the tile's last stage and the first store of its epilogue."""
    v12 = pick(ks, "gemm_kernel_v12")
    assert v12
    dis = _codeobj.disassembly(LIB, "gemm_kernel_v12")
    checked = 0
    for n, ins in dis.items():
        form = int(re.search(r"Lb[01]ELi(\d)EEEv", n).group(1))
        st = next(i for i, x in enumerate(ins) if x.startswith("buffer_store"))
        bar = max(i for i in range(st) if ins[i] == "s_barrier")
        loads = sum(1 for x in ins[bar:st] if x.startswith("buffer_load_dwordx4"))
        mfma = sum(1 for x in ins[bar:st] if x.startswith("v_mfma"))
        assert 32 <= mfma <= 40, (n, mfma)                    # the window: the 40 MFMAs behind the barrier (the first stores may overtake the last few)
        assert (loads >= 32) if form == 4 else (loads == 32), (n, loads)      # form 4 loads its per-row group bias inside the epilogue
        checked += 1
    assert checked == len(v12)


def test_the_256x320_tile_keeps_every_accumulator_where_the_source_pins_it(ks):
    """gemm_kernel_v13 (csrc/gemm_v13.h).  The 256 x 320 tile has 320 accumulators per lane; with the MFMA builtin hipcc moved ~1000 of them
    between the AGPR and the VGPR half every stage and spilled (round 1 dropped the tile for that).  The kernel writes the MFMAs as inline asm
    with the allocation class in the constraint; what must then be true of the emitted code: no scratch, no spill; 240 MFMAs (three copies of
    the 80-MFMA stage), a fifth of them on VGPR accumulators; nothing but MFMAs, LDS reads, LDS-DMA and address arithmetic between the first
    and the last MFMA — no v_accvgpr_* at all; and, because the hazard recogniser does not see inline-asm MFMAs, the wait states of acc_fence
    directly behind the last MFMA and in front of the first one."""
    v13 = pick(ks, "gemm_kernel_v13")
    assert v13
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


def test_the_row_major_v_attention_reads_v_through_the_transposing_lds_read(ks):
    """attn_fwd_kernel7 (csrc/attn_v7.h): the V tile staged row-major and transposed on the way out of LDS.  It reads every V^T fragment with two
    ds_read_b64_tr_b16 — 16 per key tile, in each of the three copies of the tile body (first tile, steady state, ragged tail) — where a V^T image
    took eight ds_read_b128; the K fragments stay on ds_read_b128; no 16-register copies of the reference-maximum splat in front of the tile's MFMAs.
    Round 6: per tile body 32 algorithmic 32 x 32 x 16 MFMAs + 8 16 x 16 x 32 ones for the softmax denominator (was 40 of the large shape)."""
    v7 = pick(ks, "attn_fwd_kernel7")
    assert len(v7) == 2                                  # f16 / bf16
    for n, ins in _codeobj.disassembly(LIB, "attn_fwd_kernel7").items():
        assert ins.count("ds_read_b64_tr_b16") == 48 and ins.count("ds_read_b128") == 24, (n, ins.count("ds_read_b64_tr_b16"), ins.count("ds_read_b128"))
        # 220 before the 16-byte-store epilogue; its v_permlane32_swap pairs are tied in / out operands of an asm statement: ~24 copies, once per block
        assert ins.count("v_mov_b64_e32") <= 64 and ins.count("v_mov_b32_e32") <= 260, (n, ins.count("v_mov_b64_e32"), ins.count("v_mov_b32_e32"))
        big = sum(1 for x in ins if x.startswith("v_mfma_f32_32x32x16"))
        small = sum(1 for x in ins if x.startswith("v_mfma_f32_16x16x32"))
        assert (big, small) == (96, 24), (n, big, small)


def test_the_fp32_convolution_spreads_its_lds_dma_between_the_mfmas(ks):
    """conv_f32_kernel (csrc/gemm_f32.hip, round 6): two blocks per CU (<= 128 VGPRs, no scratch), and inside the K loop never two LDS-DMA pieces
    back to back — each of the stage's eight sits behind an MFMA (a burst cost 70 cycles of MFMA issue per piece, the spread placement 39:
    tools/ubench/mfma_f32_rate.hip; 123 -> 141 TF/s on the VAE's up blocks) — with no 64-bit address arithmetic between them."""
    inst = pick(ks, "conv_f32_kernel")
    assert len(inst) == 1
    for n, k in inst.items():
        assert k["vgpr_count"] + k.get("agpr_count", 0) <= 128 and k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, n
    for n, ins in _codeobj.disassembly(LIB, "conv_f32_kernel").items():
        last_barrier = max(i for i, m in enumerate(ins) if m == "s_barrier")
        end = next(i for i in range(last_barrier, len(ins)) if ins[i].startswith("s_cbranch"))
        loop = ins[last_barrier:end]
        assert loop.count("v_mfma_f32_32x32x2_f32") == 64 and loop.count("buffer_load_dwordx4") == 8, (n, loop.count("v_mfma_f32_32x32x2_f32"))
        dma = [i for i, m in enumerate(loop) if m == "buffer_load_dwordx4"]
        for a, b in zip(dma, dma[1:]):
            assert loop[a:b].count("v_mfma_f32_32x32x2_f32") >= 4, (n, loop[a:b])
        assert not any(m.startswith(("v_lshl_add_u64", "v_addc", "v_mad_u64", "v_mul_lo", "v_cndmask")) for m in loop), n


def test_conv_out_pixel_kernel_takes_its_weights_through_the_scalar_cache(ks):
    """conv_out_pixel_kernel (csrc/misc.hip, round 5): the design is that the wave-uniform weights never touch the vector memory path — scalar loads
    feeding v_dot2c as its SGPR operand, 64 of them per four 16-byte loads of the lane's own pixel — at full occupancy (34 VGPRs, no scratch)."""
    inst = pick(ks, "conv_out_pixel_kernel")
    assert len(inst) == 2, list(inst)                    # f16 / bf16
    for n, k in inst.items():
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0 and k["vgpr_count"] <= 64, (n, k["vgpr_count"])
    for n, ins in _codeobj.disassembly(LIB, "conv_out_pixel_kernel").items():
        dots = sum(1 for x in ins if x.startswith("v_dot2c_f32_"))
        assert dots == 64, (n, dots)                                                      # 4 vectors x 4 output channels x 4 dwords
        assert sum(1 for x in ins if x.startswith("global_load_dwordx4")) == 4, n         # the pixel's channels only: no vector load of a weight
        assert any(x.startswith("s_load_dwordx16") for x in ins), n                        # 64 contiguous weight bytes per output channel


def test_the_generated_k_loop_schedule_is_what_the_generator_emits(tmp_path):
    """omg_amd/csrc/gemm_v12_sched.inc is GENERATED (tools/gen_ksched.py) and committed so that hipcc needs no python at build time: the committed file
    must be what the generator writes today."""
    import importlib.util
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "repo"
    (fake / "tools").mkdir(parents=True)
    (fake / "omg_amd" / "csrc").mkdir(parents=True)
    shutil.copy(os.path.join(root, "tools", "gen_ksched.py"), fake / "tools" / "gen_ksched.py")
    spec = importlib.util.spec_from_file_location("gen_ksched_copy", str(fake / "tools" / "gen_ksched.py"))
    import subprocess
    import sys
    subprocess.run([sys.executable, str(fake / "tools" / "gen_ksched.py")], check=True, capture_output=True)
    assert (fake / "omg_amd" / "csrc" / "gemm_v12_sched.inc").read_text() == open(os.path.join(root, "omg_amd", "csrc", "gemm_v12_sched.inc")).read()
    assert spec is not None
