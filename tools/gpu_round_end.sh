#!/bin/bash
# full GPU suite + headline bench (round-end rehearsal)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r03_pytest_gpu.log 2>&1
tail -25 gpurun_out/r03_pytest_gpu.log
python bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} > gpurun_out/r03_bench_fp16_${TAG:-v2}.json 2> gpurun_out/r03_bench_fp16_${TAG:-v2}.err
head -c 600 gpurun_out/r03_bench_fp16_${TAG:-v2}.json
