#!/bin/bash
# round-3 final measurement set (everything lands in gpurun_out/, copied to profiles/ afterwards)
mkdir -p gpurun_out
python bench.py --dtype fp8 --steps 3 --warmup 1 --by-shape gpurun_out/r03_by_shape_fp8_v1.txt > gpurun_out/r03_bench_fp8_v1.json 2> gpurun_out/r03_bench_fp8_v1.err
tail -c 600 gpurun_out/r03_bench_fp8_v1.json | head -c 300; echo
python bench.py --steps 3 --warmup 1 --by-shape gpurun_out/r03_by_shape_fp16_v2.txt > gpurun_out/r03_bench_fp16_v3.json 2> gpurun_out/r03_bench_fp16_v3.err
head -c 400 gpurun_out/r03_bench_fp16_v3.json; echo
python tools/bench_extras.py config4 --dtype fp16 > gpurun_out/r03_bench_extras.jsonl 2> gpurun_out/r03_bench_extras.err
python tools/bench_extras.py config4 --dtype fp8 >> gpurun_out/r03_bench_extras.jsonl 2>> gpurun_out/r03_bench_extras.err
cat gpurun_out/r03_bench_extras.jsonl | cut -c 1-220
timeout 300 python tools/vs_hipblaslt.py --rounds 3 > gpurun_out/r03_vs_hipblaslt_v3.log 2>&1
tail -11 gpurun_out/r03_vs_hipblaslt_v3.log
python tools/attn_bench.py > gpurun_out/r03_attn_bench.log 2>&1; tail -7 gpurun_out/r03_attn_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_rocprof -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph --dedup-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r03_rocprof_bench.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/r03_rocprof | head; 
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r03_rocprof -name "*kernel_stats.csv" | head -1); head -12 "$f"
# keep only the stats (the trace itself is hundreds of MB)
find $GRAFT_REPO_ROOT/gpurun_out/r03_rocprof -name "*kernel_trace.csv" -delete
find $GRAFT_REPO_ROOT/gpurun_out/r03_rocprof -name "*.db" -delete
