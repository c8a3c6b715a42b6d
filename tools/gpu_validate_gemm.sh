#!/bin/bash
# After a change to the 16-bit GEMM / conv kernels: the kernel-level parity tests, the bench-shape parity tests and the full-size identities,
# then the library comparison and the per-CU timeline of two bench shapes.   gpurun --timeout 900 -- 'bash tools/gpu_validate_gemm.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${ROUND:-r05}
python -m pytest tests/test_kernels_gpu.py tests/test_benchsize_parity_gpu.py tests/test_unet_gpu.py -x -q 2>&1 | tail -6 | tee gpurun_out/${ROUND:-r05}/validate_gemm_tests.log
python -m pytest tests/test_fullsize_properties_gpu.py -x -q -k "not unet and not oracle" 2>&1 | tail -4 | tee -a gpurun_out/${ROUND:-r05}/validate_gemm_tests.log
timeout 200 python tools/vs_hipblaslt.py --rounds 3 2>&1 | grep -v libdrm | tee gpurun_out/${ROUND:-r05}/vs_hipblaslt_v0.log
(timeout 120 python tools/gemm_timeline.py 65536 10240 1280 0 15; timeout 120 python tools/gemm_timeline.py 262144 640 640 0 28; timeout 120 python tools/gemm_timeline.py 65536 1280 5120 0 15) 2>&1 | grep -v libdrm | tee gpurun_out/${ROUND:-r05}/gemm_timeline.log
