#!/bin/bash
# Round 5's FIRST GPU call: the three experiments round 4 prepared without GPU time, on ONE box (boxes of the pool differ by up to 5 %: the A/Bs
# and the four benchmark lines must share one).  In the build container first:
#     make -C omg_amd/csrc EXP=1 DEV=1            # fp16 kernels only; the .so travels with the snapshot
#     bash tools/exp/build_alt.sh gelu2 GELU2=1 DEV=1      # the second library of tools/gpu_exp_gelu2.sh
#     gpurun --timeout 2400 -- 'bash tools/gpu_round5_first.sh'
#     make -C omg_amd/csrc clean && make -C omg_amd/csrc     # back to the product build afterwards
# Order = cheapest decisive answer first; every numerics block stops ITS experiment on a failure, not the others.
#   1. tr16 probe + attn_fwd_kernel7 bitwise vs v3 + attention microbenchmark          (~2 min)
#   2. gemm_kernel_v13 (256 x 320 tile) bitwise + A/B on convolutions and Linears        (~4 min)
#   3. gemm_kernel_v12 (window behind the last barrier, persistent) bitwise + A/B        (~4 min)
#   4. whole-benchmark lines, two steps each: product heuristic, v13 where it removes padding (29), v13 wherever N % 320 == 0 (30),
#      row-major-V attention — skipped for an experiment whose numerics failed                                     (~10 min)
# Everything lands in gpurun_out/r05/ (copy what is kept into profiles/r05_*).
cd $GRAFT_REPO_ROOT
BENCH=0 bash tools/gpu_exp_attn_v7.sh; ATTN_RC=$?
BENCH=0 bash tools/gpu_exp_v13.sh; V13_RC=$?
bash tools/gpu_exp_v12.sh; V12_RC=$?
# the short way to the first LDS-DMA (gemm_v11.h SCH == 10, variant 31: launch parameters in one batch, adapter id through the scalar cache — the
# product prologue makes six serialised scalar-cache round trips and, with weight slots, a vector-memory one, per tile): bitwise, then A/B on the
# transformer Linears as the fused steps launch them (64 weight slots) and on the plain shapes
OMG_EXP_ONLY=31 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "(variants_are_bitwise and dtype0) or 256x320_tile" 2>&1 | tail -3 | tee gpurun_out/r05/exp_v31_test.log
if grep -q passed gpurun_out/r05/exp_v31_test.log && ! grep -q failed gpurun_out/r05/exp_v31_test.log; then
  timeout 300 python tools/ksched_ab.py 25,31,28,32 3 slots 2>&1 | grep -v libdrm | tee gpurun_out/r05/exp_v31_ab_slots.log      # 32 = the 256 x 320 tile (28) with the same prologue
  timeout 300 python tools/ksched_ab.py 25,31 3 k 2>&1 | grep -v libdrm | tee gpurun_out/r05/exp_v31_ab_k.log
fi
# conv_out with the weight slice in registers (tools/exp/conv_out_v2.h; 0.45 % of the step at 14x its memory time): bitwise, then the two kernels timed
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_out_with_the_weight_slice" 2>&1 | tail -3 | tee gpurun_out/r05/exp_conv_out_test.log
timeout 120 python tools/exp/conv_out_bench.py 2>&1 | grep -v libdrm | tee gpurun_out/r05/exp_conv_out_bench.log
# the one-transcendental GELU (tools/exp/gelu_v2.h), if its second library was built: gate function to one ulp, GEGLU launches timed on both libraries
[ -f tools/exp/build/gelu2/libomg_hip.so ] && bash tools/gpu_exp_gelu2.sh
echo "numerics: attn_v7 rc=$ATTN_RC  gemm_v13 rc=$V13_RC  gemm_v12 rc=$V12_RC"
O=gpurun_out/r05
B="--steps 2 --warmup 1 --dedup-steps 0 --no-cpu-baseline"
timeout 600 python bench.py $B --by-shape $O/first_by_shape_v0.txt > $O/first_bench_v0.json 2> $O/first_bench_v0.err; head -c 300 $O/first_bench_v0.json; echo
if [ $V13_RC = 0 ]; then
  for V in 29 30; do
    OMG_GEMM_VARIANT=$V timeout 600 python bench.py $B --by-shape $O/first_by_shape_v$V.txt > $O/first_bench_v$V.json 2> $O/first_bench_v$V.err; head -c 300 $O/first_bench_v$V.json; echo
  done
fi
if [ $ATTN_RC = 0 ]; then
  timeout 600 python tools/exp/run_patched.py bench.py $B --by-shape $O/first_by_shape_attn_v7.txt > $O/first_bench_attn_v7.json 2> $O/first_bench_attn_v7.err; head -c 300 $O/first_bench_attn_v7.json; echo
fi
