set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
( timeout 600 python tools/v9_check.py 2 ) > gpurun_out/r03/v9_check.log 2>&1
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "variants or gemm or conv" 2>&1 | tail -15 ) > gpurun_out/r03/tests3a.log 2>&1
( timeout 1500 python -m pytest tests/test_compat_instantid_gpu.py tests/test_pipeline_gpu.py tests/test_fullsize_properties_gpu.py tests/test_compat_gpu.py tests/test_mx8_gpu.py tests/test_config4_gpu.py -q -s -x 2>&1 | tail -60 ) > gpurun_out/r03/tests3.log 2>&1
cat gpurun_out/r03/v9_check.log; tail -15 gpurun_out/r03/tests3a.log; tail -40 gpurun_out/r03/tests3.log
