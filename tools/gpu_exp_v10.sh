#!/bin/bash
# Next step for the K-loop experiment (omg_amd/csrc/gemm_v10_exp.h).  In the build container:
#     make -C omg_amd/csrc EXP=1 -j8            # the .so travels to the GPU box with the snapshot
#     gpurun --timeout 600 -- 'bash tools/gpu_exp_v10.sh'
#     make -C omg_amd/csrc clean && make -C omg_amd/csrc -j8     # back to the product build afterwards
# 1. numerics first: variant 35 must be torch.equal with variant 1 on every epilogue form (the test adds 35 when the kernel is in the library)
# 2. then the interleaved A/B against v7-XE (25) on the benchmark's shapes, K = 1280 / 5120 and the N = 640 family
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "variants_are_bitwise" 2>&1 | tail -5 | tee gpurun_out/exp_v10_test.log
grep -q passed gpurun_out/exp_v10_test.log || exit 1
timeout 400 python tools/ksched_ab.py 25,35 5 k 2>&1 | grep -v libdrm | tee gpurun_out/exp_v10_ab_k.log
timeout 400 python tools/ksched_ab.py 25,35 5 n640 2>&1 | grep -v libdrm | tee gpurun_out/exp_v10_ab_n640.log
