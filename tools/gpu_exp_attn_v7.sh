#!/bin/bash
# Round 5, third experiment prepared (compiled, inspected, never run) in round 4 — attn_fwd_kernel7 (tools/exp/attn_v7.h): the self-attention
# kernel reading V row-major through ds_read_b64_tr_b16, which makes omg_transpose_v (0.9 % of the step) unnecessary.  In the build container
# (the same EXP build serves gpu_exp_v12.sh / gpu_exp_v13.sh):
#     make -C omg_amd/csrc EXP=1 DEV=1
#     gpurun --timeout 600 -- 'bash tools/gpu_exp_attn_v7.sh'
# 1. the instruction's semantics on the part (tools/exp/tr16_probe.hip: three PASS lines, a few seconds) — the kernel's addressing is built on them;
# 2. bitwise against v3 (plain, borrowed Q / K, accumulate; whole and ragged tiles);  3. the attention microbenchmark, v3 | v7, with the
# transpose_v time the new kernel saves printed beside it.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
hipcc --offload-arch=gfx950 -O2 tools/exp/tr16_probe.hip -o gpurun_out/tr16_probe 2> $O/exp_tr16_probe_build.log && timeout 60 gpurun_out/tr16_probe 2>&1 | tee $O/exp_tr16_probe.log
grep -c PASS $O/exp_tr16_probe.log | grep -q 3 || { echo "tr16 semantics differ from the guide's statement: see $O/exp_tr16_probe.log"; exit 1; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "row_major_v_attention" 2>&1 | tail -8 | tee $O/exp_attn_v7_test.log
grep -q passed $O/exp_attn_v7_test.log || exit 1
grep -q failed $O/exp_attn_v7_test.log && exit 1
timeout 300 python tools/attn_bench.py 3 7 8 2>&1 | grep -v libdrm | tee $O/exp_attn_v7_bench.log
# the lockstep question (header of tools/exp/attn_v7.h): half of the first-round workgroups started 10 / 20 / 40 us late (variant 9 = 8 + the knobs)
timeout 300 python tools/attn_bench.py 8 0xa09 0x1409 0x2809 2>&1 | grep -v libdrm | tee $O/exp_attn_v7_stagger.log
# ... and the XCD-aware block order (bit 16: the query blocks of one (sample, head) on ONE XCD's L2 instead of up to eight), against variant 8 and the knob-carrying kernel without it
timeout 300 python tools/attn_bench.py 8 9 0x10009 2>&1 | grep -v libdrm | tee $O/exp_attn_v7_xcd.log
# 4. the whole benchmark with the self-attention calls on kernel 7 (tools/exp/rowmajor_v_patch.py: the product's Python unchanged, transpose_v skipped);
#    the baseline line on the same box is gpu_exp_v13.sh's variant-0 run (tools/gpu_round5_first.sh runs both in one call)
if [ "${BENCH:-1}" = 1 ]; then
  timeout 600 python tools/exp/run_patched.py bench.py --steps 2 --warmup 1 --dedup-steps 0 --no-cpu-baseline --by-shape $O/exp_attn_v7_by_shape.txt > $O/exp_attn_v7_bench.json 2> $O/exp_attn_v7_bench.err
  head -c 400 $O/exp_attn_v7_bench.json; echo
fi
