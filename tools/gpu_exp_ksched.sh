#!/bin/bash
# K-loop schedule experiment (omg_amd/csrc/gemm_v11.h, tools/gen_ksched.py).  In the build container:
#     make -C omg_amd/csrc EXP=1 DEV=1          # the .so travels to the GPU box with the snapshot (DEV: fp16 kernels only)
#     gpurun --timeout 600 -- 'bash tools/gpu_exp_ksched.sh'
#     make -C omg_amd/csrc clean && make -C omg_amd/csrc     # back to the product build afterwards
# 1. numerics first: variants 35..39 must be torch.equal with variant 1 on every epilogue form
# 2. then the interleaved A/B against v7-XE (25) on the benchmark's Linear shapes, and the library on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_kernels_gpu.py -x -q -k "variants_are_bitwise and dtype0" 2>&1 | tail -5 | tee gpurun_out/r04/exp_ksched_test.log
grep -q passed gpurun_out/r04/exp_ksched_test.log || exit 1
timeout 300 python tools/ksched_ab.py 25,40,43,44 3 k 2>&1 | grep -v libdrm | tee gpurun_out/r04/exp_ksched_ab_k.log
timeout 300 python tools/ksched_ab.py 25,36,40,43,44 3 conv 2>&1 | grep -v libdrm | tee gpurun_out/r04/exp_ksched_ab_conv.log

