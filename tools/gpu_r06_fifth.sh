#!/bin/bash
# round 6, fifth GPU call: the persistent MX-fp8 GEMM (gemm_mx8_kernel_p) — bitwise against the one-tile form, the whole mx8 suite, the microbenchmark with
# both forms side by side, and the fp8 bench line with each (one box, interleaved)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_mx8_gpu.py -x -q --durations=5 > $O/fifth_mx8_tests.log 2>&1
tail -12 $O/fifth_mx8_tests.log
timeout 600 python tools/mx8_bench.py 2>&1 | grep -v libdrm | tee $O/mx8_bench.log | tail -12
B="--steps 2 --warmup 1 --dtype fp8 --no-cpu-baseline --dedup-steps 0"
OMG_MX8_SPLIT=0x800c python bench.py $B --by-shape $O/by_shape_fp8_one_tile.txt > $O/bench_fp8_one_tile.json 2> $O/bench_fp8_one_tile.err
python bench.py $B --by-shape $O/by_shape_fp8_persistent.txt > $O/bench_fp8_persistent.json 2> $O/bench_fp8_persistent.err
OMG_MX8_SPLIT=0x800c python bench.py $B --no-roofline > $O/bench_fp8_one_tile_b.json 2> $O/bench_fp8_one_tile_b.err
python bench.py $B --no-roofline > $O/bench_fp8_persistent_b.json 2> $O/bench_fp8_persistent_b.err
for f in one_tile persistent one_tile_b persistent_b; do python -c "import json;d=json.load(open('$O/bench_fp8_$f.json'));print('$f', d['value'], (d.get('roofline') or {}).get('achieved'))"; done
head -12 $O/by_shape_fp8_one_tile.txt; head -12 $O/by_shape_fp8_persistent.txt
