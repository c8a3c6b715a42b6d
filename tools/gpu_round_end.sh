#!/bin/bash
# full GPU suite + headline bench (round-end rehearsal)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r03_pytest_gpu.log 2>&1
tail -25 gpurun_out/r03_pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_fp16_v2.json 2> gpurun_out/r03_bench_fp16_v2.err
tail -c 3000 gpurun_out/r03_bench_fp16_v2.json
