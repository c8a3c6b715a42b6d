set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
( timeout 300 python tools/vs_hipblaslt.py --rounds 3 ) > gpurun_out/r03/vs_hipblaslt_v2.log 2>&1
( timeout 300 python tools/xe_time.py ) > gpurun_out/r03/xe_time_v2.log 2>&1
( timeout 120 python tools/gemm_timeline.py 65536 10240 1280 0 25; timeout 120 python tools/gemm_timeline.py 65536 1280 1280 0 25 ) > gpurun_out/r03/timeline3.log 2>&1
( timeout 300 python tools/mx8_bench.py ) > gpurun_out/r03/mx8_bench_v2.log 2>&1
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_mx8_gpu.py tests/test_benchsize_parity_gpu.py tests/test_pipeline_gpu.py tests/test_config4_gpu.py -q -x 2>&1 | tail -15 ) > gpurun_out/r03/tests6.log 2>&1
cat gpurun_out/r03/vs_hipblaslt_v2.log gpurun_out/r03/xe_time_v2.log gpurun_out/r03/timeline3.log; tail -30 gpurun_out/r03/mx8_bench_v2.log; tail -15 gpurun_out/r03/tests6.log
