#!/bin/bash
# Round 5, fourth GPU call: the whole GPU suite on the landed kernels (GELU landed, ABI 6), the headline line, the attention block order in situ,
# the tile-walk group size (fabric traffic experiment)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 > $O/fourth_pytest_gpu.log 2>&1
tail -16 $O/fourth_pytest_gpu.log
B="--steps 2 --warmup 1 --dedup-steps 0 --no-cpu-baseline"
timeout 600 python bench.py $B --by-shape $O/fourth_by_shape.txt > $O/fourth_bench.json 2> $O/fourth_bench.err; head -c 300 $O/fourth_bench.json; echo
OMG_ATTN_VARIANT=0x10000 timeout 600 python bench.py $B --by-shape $O/fourth_by_shape_natural.txt > $O/fourth_bench_natural.json 2> $O/fourth_bench_natural.err; head -c 300 $O/fourth_bench_natural.json; echo
timeout 300 python tools/ksched_ab.py 25,16409,32793 3 k 2>&1 | grep -v libdrm | tee $O/fourth_walk_group_ab.log
export TMPDIR=/tmp
timeout 300 python tools/pmc_traffic.py fp16 $O/fourth_pmc_traffic_g8.json 25 2>&1 | tail -3
timeout 300 python tools/pmc_traffic.py fp16 $O/fourth_pmc_traffic_g4.json 16409 2>&1 | tail -3
