set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
( timeout 900 python bench.py --steps 2 --warmup 1 --by-shape gpurun_out/r03/r03_by_shape_fp16_v1.txt ) > gpurun_out/r03/r03_bench_fp16_v1.json 2> gpurun_out/r03/bench_fp16_v1.err
tail -3 gpurun_out/r03/bench_fp16_v1.err; cat gpurun_out/r03/r03_bench_fp16_v1.json | cut -c1-1500
