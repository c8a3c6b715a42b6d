#!/bin/bash
# round 6, third GPU call: attention with the 16x16x32 denominator as the default + the prologue reorder (A/B against the Q-first build), the MFMA-shape
# probe (32x32x16 vs 16x16x32 on the GEMM's K-loop mix with power / clock), in-situ A/B of the denominator form on the whole benchmark step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_properties_gpu.py -x -q -k "attention or non_finite or row_major or denominator" > $O/third_tests.log 2>&1
tail -6 $O/third_tests.log; grep "gate = " $O/third_tests.log
for r in 1 2; do
  echo "== Q-first prologue (alt build), DEN 1"; OMG_HIP_LIB=$GRAFT_REPO_ROOT/omg_amd/csrc/libomg_hip_qfirst.so timeout 300 python tools/attn_bench.py 0 2>&1 | grep "^(" | head -2
  echo "== DMA-first prologue (this build), DEN 1"; timeout 300 python tools/attn_bench.py 0 2>&1 | grep "^(" | head -2
done 2>&1 | tee $O/attn_prologue_ab.log
timeout 300 python tools/mfma_shape_probe.py 3 2>&1 | grep -v libdrm | tee $O/mfma_shape_probe.log
OMG_ATTN_VARIANT=7 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dedup-steps 0 --by-shape $O/by_shape_fp16_den0.txt > $O/bench_fp16_den0.json 2> $O/bench_fp16_den0.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dedup-steps 0 --by-shape $O/by_shape_fp16_den1.txt > $O/bench_fp16_den1.json 2> $O/bench_fp16_den1.err
OMG_ATTN_VARIANT=7 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dedup-steps 0 --no-roofline > $O/bench_fp16_den0b.json 2> $O/bench_fp16_den0b.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dedup-steps 0 --no-roofline > $O/bench_fp16_den1b.json 2> $O/bench_fp16_den1b.err
for f in den0 den1 den0b den1b; do python -c "import json;d=json.load(open('$O/bench_fp16_$f.json'));print('$f', d['value'], d.get('roofline',{}).get('attn_kernel'))"; done
grep attn $O/by_shape_fp16_den0.txt | head -4; grep attn $O/by_shape_fp16_den1.txt | head -4
