#!/bin/bash
# full GPU suite + headline bench (round-end rehearsal):   gpurun --timeout 3600 -- 'bash tools/gpu_round_end.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/${ROUND:-r05}      # ROUND=r04 reproduces the names of profiles/r04_*
mkdir -p $O
OMG_RUN_SLOW=1 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1
tail -22 $O/pytest_gpu.log
python bench.py --gpus 1 --steps ${STEPS:-3} --warmup ${WARMUP:-1} --by-shape $O/by_shape_fp16_${TAG:-v1}.txt > $O/bench_fp16_${TAG:-v1}.json 2> $O/bench_fp16_${TAG:-v1}.err
head -c 900 $O/bench_fp16_${TAG:-v1}.json; echo
python bench.py --gpus 1 --steps 2 --warmup 1 --dtype fp8 --no-cpu-baseline --by-shape $O/by_shape_fp8_${TAG:-v1}.txt > $O/bench_fp8_${TAG:-v1}.json 2> $O/bench_fp8_${TAG:-v1}.err
head -c 500 $O/bench_fp8_${TAG:-v1}.json; echo
