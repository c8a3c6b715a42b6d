#!/bin/bash
# round 6, second GPU call: the ControlNet modes' loop tests, the attention denominator forms (bitwise-ish agreement + interleaved A/B), the baseline bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests/test_controlnet_modes_gpu.py tests/test_kernels_gpu.py -x -q -k "controlnet or denominator or non_finite or row_major or attention" --durations=5 > $O/second_tests.log 2>&1
tail -15 $O/second_tests.log
timeout 600 python tools/attn_bench.py 7 8 9 7 8 9 2>&1 | grep -v libdrm | tee $O/attn_bench_den_forms.log | tail -9
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dedup-steps 0 --by-shape $O/by_shape_fp16_v0.txt > $O/bench_fp16_v0.json 2> $O/bench_fp16_v0.err
head -c 600 $O/bench_fp16_v0.json; echo
head -30 $O/by_shape_fp16_v0.txt
