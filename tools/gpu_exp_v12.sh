#!/bin/bash
# Round 5's first GPU call: the experiment prepared (compiled, inspected, never run) at the end of round 4 — gemm_kernel_v12
# (tools/exp/gemm_v12.h): v11's ring K loop with the window behind a tile's LAST barrier used for the residual DMA (45), a persistent
# tile walk (46), the next tile's first two stages issued in front of the epilogue's stores (47) and a counted wait (48).  In the build container:
#     make -C omg_amd/csrc EXP=1 DEV=1          # the .so travels with the snapshot (DEV: fp16 kernels only)
#     gpurun --timeout 900 -- 'bash tools/gpu_exp_v12.sh'
#     make -C omg_amd/csrc clean && make -C omg_amd/csrc     # back to the product build afterwards
# 1. numerics first: every EXP variant torch.equal with variant 1 on every epilogue form, and the persistent forms walking 3-4 tiles per block
# 2. then the interleaved A/B against the product kernel (25) on the benchmark's Linear shapes, the 640-wide ones and the convolutions
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
OMG_EXP_ONLY=45,46,47,48 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "(variants_are_bitwise and dtype0) or persistent_gemm_walks" 2>&1 | tail -8 | tee $O/exp_v12_test.log
grep -q passed $O/exp_v12_test.log || exit 1
grep -q failed $O/exp_v12_test.log && exit 1
timeout 300 python tools/ksched_ab.py 25,45,46,47,48 3 k 2>&1 | grep -v libdrm | tee $O/exp_v12_ab_k.log
timeout 300 python tools/ksched_ab.py 25,45,47,48 3 n640 2>&1 | grep -v libdrm | tee $O/exp_v12_ab_n640.log
timeout 300 python tools/ksched_ab.py 25,46,47 3 conv 2>&1 | grep -v libdrm | tee $O/exp_v12_ab_conv.log
