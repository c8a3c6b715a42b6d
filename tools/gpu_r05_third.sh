#!/bin/bash
# Round 5, third GPU call: second call's failures fixed (Nkv_pad check on the row-major-V path; the packed GELU's bit_cast miscompile)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -6 | tee $O/third_kernels_test.log
timeout 300 python tools/attn_bench.py 3 7 2>&1 | grep -v libdrm | tee $O/third_attn_bench.log
ALT=$PWD/tools/exp/build/gelu2/libomg_hip.so
if [ -f $ALT ]; then
  OMG_HIP_LIB=$ALT timeout 120 python tools/exp/gelu2_diag.py 2>&1 | grep -v libdrm | tee $O/third_gelu2_diag.log
  OMG_HIP_LIB=$ALT OMG_TEST_GELU_ULP=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "one_half_precision_ulp or gemm_geglu or variants_are_bitwise or persistent_gemm or 256x320" 2>&1 | tail -4 | tee $O/third_gelu2_test.log
  for L_ in product gelu2; do
    [ $L_ = gelu2 ] && export OMG_HIP_LIB=$ALT
    timeout 300 python tools/ksched_ab.py 25 3 geglu 2>&1 | grep -v libdrm | sed "s/^/$L_  /" | tee -a $O/third_gelu2_ab.log
  done
  unset OMG_HIP_LIB
fi
B="--steps 2 --warmup 1 --dedup-steps 0 --no-cpu-baseline"
timeout 600 python bench.py $B --by-shape $O/third_by_shape.txt > $O/third_bench.json 2> $O/third_bench.err; head -c 400 $O/third_bench.json; echo
