#!/bin/bash
# Round 5, seventh GPU call: attn_fwd_kernel7 with running K / V row pointers (no 64-bit multiplies in the tile loop)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or attn or row_major" 2>&1 | tail -3 | tee $O/seventh_attn_tests.log
timeout 300 python tools/attn_bench.py 3 7 2>&1 | grep -v libdrm | tee $O/seventh_attn_bench.log
export TMPDIR=/tmp
timeout 300 python tools/pmc_attention.py $O/seventh_pmc_attention.json 2>&1 | tail -2
B="--steps 2 --warmup 1 --dedup-steps 0 --no-cpu-baseline"
timeout 600 python bench.py $B --by-shape $O/seventh_by_shape.txt > $O/seventh_bench.json 2> $O/seventh_bench.err; head -c 300 $O/seventh_bench.json; echo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
