# interleaved A/B on ONE box: attn_fwd_kernel7 with the next tile's LDS-DMA pieces as a burst behind the barrier / spread behind the S^T MFMAs
mkdir -p gpurun_out/r06
for r in 1 2 3; do
  for v in burst spread; do
    echo "== $v"
    OMG_HIP_LIB=$PWD/tools/alt/libomg_attn_$v.so python tools/attn_bench.py 0 2>&1 | grep "^(64,10,4096,4096)\|^(64,20,1024,1024)\|^(32,10,4096,4096)"
  done
done 2>&1 | tee gpurun_out/r06/attn_dma_spread_ab.log
OMG_HIP_LIB=$PWD/tools/alt/libomg_attn_spread.so python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention or attn" 2>&1 | tail -2
