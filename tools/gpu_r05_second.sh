#!/bin/bash
# Round 5, second GPU call: the landed kernels (gemm_kernel_v12 = variant 25, gemm_kernel_v13 = variant 28, attn_fwd_kernel7) as the product.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -6 | tee $O/second_kernels_test.log
timeout 300 python tools/attn_bench.py 3 7 2>&1 | grep -v libdrm | tee $O/second_attn_bench.log
# GELU2: which variants disagree on the alt library, then the GEGLU launches on both
ALT=$PWD/tools/exp/build/gelu2/libomg_hip.so
if [ -f $ALT ]; then
  OMG_HIP_LIB=$ALT timeout 120 python tools/exp/gelu2_diag.py 2>&1 | grep -v libdrm | tee $O/second_gelu2_diag.log
  for L_ in product gelu2; do
    [ $L_ = gelu2 ] && export OMG_HIP_LIB=$ALT
    timeout 300 python tools/ksched_ab.py 25 3 geglu 2>&1 | grep -v libdrm | sed "s/^/$L_  /" | tee -a $O/second_gelu2_ab.log
  done
  unset OMG_HIP_LIB
fi
B="--steps 2 --warmup 1 --dedup-steps 0 --no-cpu-baseline"
timeout 600 python bench.py $B --by-shape $O/second_by_shape.txt > $O/second_bench.json 2> $O/second_bench.err; head -c 400 $O/second_bench.json; echo
timeout 400 python tools/vs_hipblaslt.py 2>&1 | grep -v libdrm | tee $O/second_vs_hipblaslt.log
