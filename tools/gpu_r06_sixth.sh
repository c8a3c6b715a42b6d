#!/bin/bash
# round 6: gemm_kernel_v12 form 6 (persistent residual form: half the residual staged, half register-direct, next tile's stage 0 prefetched) — bitwise tests,
# interleaved microbenchmark A/B against form 2, in-situ A/B on the whole benchmark step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_benchsize_parity_gpu.py -x -q -k "gemm or conv or tile" > $O/sixth_tests.log 2>&1
tail -5 $O/sixth_tests.log
timeout 500 python tools/ksched_ab.py 33554457,25 5 n320 2>&1 | grep -v libdrm | tee $O/res_form6_ab_n320.log
timeout 400 python tools/ksched_ab.py 33554457,25 5 slots 2>&1 | grep -v libdrm | tee $O/res_form6_ab_slots.log
B="--steps 2 --warmup 1 --no-cpu-baseline --dedup-steps 0"
OMG_GEMM_VARIANT=33554432 python bench.py $B --by-shape $O/by_shape_fp16_form2.txt > $O/bench_fp16_form2.json 2> $O/bench_fp16_form2.err
python bench.py $B --by-shape $O/by_shape_fp16_form6.txt > $O/bench_fp16_form6.json 2> $O/bench_fp16_form6.err
OMG_GEMM_VARIANT=33554432 python bench.py $B --no-roofline > $O/bench_fp16_form2b.json 2> $O/bench_fp16_form2b.err
python bench.py $B --no-roofline > $O/bench_fp16_form6b.json 2> $O/bench_fp16_form6b.err
for f in form2 form6 form2b form6b; do python -c "import json;d=json.load(open('$O/bench_fp16_$f.json'));print('$f', d['value'], (d.get('roofline') or {}).get('achieved'))"; done
grep "lin', 65536, 1280, 1280\|lin', 65536, 1280, 5120\|lin', 32768, 1280" $O/by_shape_fp16_form2.txt $O/by_shape_fp16_form6.txt
