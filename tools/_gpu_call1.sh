set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
( timeout 1500 python -m pytest tests/test_config4_gpu.py tests/test_compat_instantid_gpu.py tests/test_pipeline_gpu.py tests/test_fullsize_properties_gpu.py tests/test_compat_gpu.py tests/test_mx8_gpu.py -q -s -x 2>&1 | tail -60 ) > gpurun_out/r03/tests1.log 2>&1
( timeout 300 python tools/vs_hipblaslt.py --rounds 3 ) > gpurun_out/r03/vs_hipblaslt.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03/hbl_names --output-format csv -- python $GRAFT_REPO_ROOT/tools/vs_hipblaslt.py --names ) > gpurun_out/r03/hbl_names.log 2>&1
( timeout 400 python tools/pmc_traffic.py fp16 gpurun_out/r03/r03_pmc_traffic_fp16.json ) > gpurun_out/r03/pmc_fp16.log 2>&1
( timeout 400 python tools/pmc_traffic.py fp8 gpurun_out/r03/r03_pmc_traffic_fp8.json ) > gpurun_out/r03/pmc_fp8.log 2>&1
tail -5 gpurun_out/r03/tests1.log; cat gpurun_out/r03/vs_hipblaslt.log; cat gpurun_out/r03/pmc_fp16.log | tail -3; cat gpurun_out/r03/pmc_fp8.log | tail -3
