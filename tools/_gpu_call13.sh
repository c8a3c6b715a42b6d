cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_mx8_gpu.py -q -x 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r03/tests13.log
( timeout 300 python tools/mx8_bench.py ) 2>&1 | grep -v libdrm | tee gpurun_out/r03/mx8_bench_v3.log
