#!/bin/bash
# Round 5, second experiment prepared (compiled, inspected, never run) in round 4 — gemm_kernel_v13 (tools/exp/gemm_v13.h): the 256 x 320 tile
# on four waves with class-pinned inline-asm MFMAs (variant 27: register-direct epilogue, 28: transposed streaming epilogue for the aligned
# 128-column groups).  In the build container (the same EXP build serves tools/gpu_exp_v12.sh):
#     make -C omg_amd/csrc EXP=1 DEV=1          # the .so travels with the snapshot (DEV: fp16 kernels only)
#     gpurun --timeout 900 -- 'bash tools/gpu_exp_v13.sh'
#     make -C omg_amd/csrc clean && make -C omg_amd/csrc     # back to the product build afterwards
# 1. numerics first: 27 / 28 torch.equal with variant 1 on every epilogue form they have (the variants test + the tile's own test: N = 320 k and
#    ragged widths, one / two / three-stage K, weight slots with a skipped group, convolutions with folded and per-row group bias, residual)
# 2. then the interleaved A/B: against the heuristic's choice (0: the 128 x 320 tile) on the N = 320 / 640 convolutions, against the product
#    256 x 256 kernel (25) on every Linear width that is a multiple of 320, and on the 256-tile convolutions
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
OMG_EXP_ONLY=27,28,32 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "(variants_are_bitwise and dtype0) or 256x320_tile" 2>&1 | tail -8 | tee $O/exp_v13_test.log
grep -q passed $O/exp_v13_test.log || exit 1
grep -q failed $O/exp_v13_test.log && exit 1
timeout 300 python tools/ksched_ab.py 0,27,28 3 conv320 2>&1 | grep -v libdrm | tee $O/exp_v13_ab_conv320.log
timeout 300 python tools/ksched_ab.py 25,27,28 3 n320 2>&1 | grep -v libdrm | tee $O/exp_v13_ab_n320.log
timeout 300 python tools/ksched_ab.py 25,27,28 3 conv 2>&1 | grep -v libdrm | tee $O/exp_v13_ab_conv.log
# 3. if the tile pays: the whole benchmark with the heuristic allowed to pick it (variant 29: where it removes padding or replaces the 128 x 320 tile;
#    30: wherever N is a multiple of 320), A/B on this box against the product heuristic — two steps each, no CPU leg, per-shape tables kept
if [ "${BENCH:-1}" = 1 ]; then
  for V in 0 29 30; do
    OMG_GEMM_VARIANT=$V timeout 600 python bench.py --steps 2 --warmup 1 --dedup-steps 0 --no-cpu-baseline --by-shape $O/exp_v13_by_shape_v$V.txt > $O/exp_v13_bench_v$V.json 2> $O/exp_v13_bench_v$V.err
    head -c 400 $O/exp_v13_bench_v$V.json; echo
  done
fi
