# interleaved A/B on ONE box: attn_fwd_kernel7 with K / V tiles by global_load_lds (64-bit lane addresses) / by buffer_load lds (descriptor + 32-bit lane offset + scalar tile offset)
mkdir -p gpurun_out/r06
for r in 1 2 3; do
  for v in global buffer; do
    echo "== $v"
    OMG_HIP_LIB=$PWD/tools/alt/libomg_attn_$v.so python tools/attn_bench.py 0 2>&1 | grep "^(64,10,4096,4096)\|^(64,20,1024,1024)\|^(32,10,4096,4096)"
  done
done 2>&1 | tee gpurun_out/r06/attn_dma_descriptor_ab.log
OMG_HIP_LIB=$PWD/tools/alt/libomg_attn_buffer.so python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention or attn" 2>&1 | tail -2
