#!/bin/bash
# Round 5: the GEGLU epilogue's gate function through ONE transcendental (tools/exp/gelu_v2.h).  It changes the product's arithmetic (5.7 % of the
# fp16 outputs of a GEGLU by one ulp, each form as close to the exact value as the other — tests/test_gelu_v2.py), so it is built as a SECOND
# library and judged in two steps.  In the build container:
#     bash tools/exp/build_alt.sh gelu2 GELU2=1 DEV=1              # tools/exp/build/gelu2/libomg_hip.so travels with the snapshot
#     gpurun --timeout 900 -- 'bash tools/gpu_exp_gelu2.sh'
# 1. the gate function alone to one fp16 ulp, on both libraries; the GEGLU kernel tests on the new one;  2. the GEGLU launches of the benchmark
#    timed on both libraries (same box, one process each).  If both are good: `make -C omg_amd/csrc clean && make -C omg_amd/csrc GELU2=1`, the
#    full -m gpu suite + bench.py on that build, then move the form into common.h.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
ALT=$PWD/tools/exp/build/gelu2/libomg_hip.so
[ -f $ALT ] || { echo "no $ALT: bash tools/exp/build_alt.sh gelu2 GELU2=1 DEV=1"; exit 1; }
OMG_TEST_GELU_ULP=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "one_half_precision_ulp" 2>&1 | tail -4 | tee $O/exp_gelu2_ulp_product.log
OMG_HIP_LIB=$ALT OMG_TEST_GELU_ULP=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "one_half_precision_ulp or (gemm_geglu and dtype0) or (variants_are_bitwise and dtype0)" 2>&1 | tail -6 | tee $O/exp_gelu2_test.log
grep -q passed $O/exp_gelu2_test.log || exit 1
grep -q failed $O/exp_gelu2_test.log && exit 1
for L_ in product gelu2; do
  [ $L_ = gelu2 ] && export OMG_HIP_LIB=$ALT
  timeout 300 python tools/ksched_ab.py 25 3 geglu 2>&1 | grep -v libdrm | sed "s/^/$L_  /" | tee -a $O/exp_gelu2_ab.log
done
