set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
( timeout 300 python tools/vs_hipblaslt.py --rounds 3 ) > gpurun_out/r03/vs_hipblaslt_v1.log 2>&1
( timeout 300 python tools/stagger_sweep.py ) > gpurun_out/r03/stagger.log 2>&1
( timeout 120 python tools/gemm_timeline.py 65536 10240 1280 0 25; timeout 120 python tools/gemm_timeline.py 65536 10240 1280 $((32<<16)) 25; timeout 120 python tools/gemm_timeline.py 65536 1280 1280 0 25 ) > gpurun_out/r03/timeline.log 2>&1
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_config4_gpu.py tests/test_compat_instantid_gpu.py tests/test_pipeline_gpu.py tests/test_fullsize_properties_gpu.py tests/test_compat_gpu.py tests/test_mx8_gpu.py -q -s -x 2>&1 | tail -80 ) > gpurun_out/r03/tests2.log 2>&1
cat gpurun_out/r03/vs_hipblaslt_v1.log gpurun_out/r03/stagger.log gpurun_out/r03/timeline.log; tail -30 gpurun_out/r03/tests2.log
