#!/bin/bash
# round 6: the self-attention epilogue with 16-byte stores (v_permlane32_swap pairs the two lane halves' adjacent pieces) — tests, interleaved A/B against the
# 8-byte-store build, in-situ A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_properties_gpu.py tests/test_unet_gpu.py -x -q -k "attention or row_major or unet or transformer" 2>&1 | tail -3
for r in 1 2 3; do
  echo "== 8-byte stores (alt build)"; OMG_HIP_LIB=$GRAFT_REPO_ROOT/omg_amd/csrc/libomg_hip_narrow.so timeout 300 python tools/attn_bench.py 0 2>&1 | grep "^(" | sed -n '1,2p;5p'
  echo "== 16-byte stores (this build)"; timeout 300 python tools/attn_bench.py 0 2>&1 | grep "^(" | sed -n '1,2p;5p'
done 2>&1 | tee $O/attn_wide_store_ab.log
B="--steps 2 --warmup 1 --no-cpu-baseline --dedup-steps 0 --no-roofline"
OMG_HIP_LIB=$GRAFT_REPO_ROOT/omg_amd/csrc/libomg_hip_narrow.so python bench.py $B > $O/bench_fp16_narrow.json 2> $O/bench_fp16_narrow.err
python bench.py $B > $O/bench_fp16_wide.json 2> $O/bench_fp16_wide.err
OMG_HIP_LIB=$GRAFT_REPO_ROOT/omg_amd/csrc/libomg_hip_narrow.so python bench.py $B > $O/bench_fp16_narrow_b.json 2> $O/bench_fp16_narrow_b.err
python bench.py $B > $O/bench_fp16_wide_b.json 2> $O/bench_fp16_wide_b.err
for f in narrow wide narrow_b wide_b; do python -c "import json;d=json.load(open('$O/bench_fp16_$f.json'));print('$f', d['value'])"; done
