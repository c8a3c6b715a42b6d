set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
(
for shape in "65536 10240 1280" "65536 1280 1280"; do
for v in 25 26; do
for extra in 0 1024; do
echo "=== shape $shape variant $v extra $extra"
timeout 120 python tools/gemm_timeline.py $shape $extra $v 2>&1 | grep -v libdrm
done; done; done ) > gpurun_out/r03/timeline2.log 2>&1
( timeout 1500 python -m pytest tests/test_compat_instantid_gpu.py tests/test_pipeline_gpu.py tests/test_fullsize_properties_gpu.py tests/test_compat_gpu.py tests/test_mx8_gpu.py tests/test_config4_gpu.py -q -s -x 2>&1 | tail -60 ) > gpurun_out/r03/tests4.log 2>&1
cat gpurun_out/r03/timeline2.log; tail -40 gpurun_out/r03/tests4.log
