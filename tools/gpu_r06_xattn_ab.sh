# interleaved A/B on ONE box: attn_fwd_kernel6 (cross-attention, <= 128 keys) computing every 16-key block of its padded last tile / only those with a real key
mkdir -p gpurun_out/r06
python -m pytest tests/test_kernels_gpu.py tests/test_attention_gpu.py -q -m gpu -k "attention or attn or cross or ip" 2>&1 | tail -2
for r in 1 2 3; do
  for v in base skip; do
    if [ $v = base ]; then lib=tools/alt/libomg_base.so; else lib=omg_amd/csrc/libomg_hip.so; fi
    echo "== $v"
    OMG_HIP_LIB=$PWD/$lib python tools/attn_bench.py 0 2>&1 | grep "^(64,20,1024,77)\|^(64,10,4096,77)\|^(64,20,1024,16)" | cut -c1-90
  done
done 2>&1 | tee gpurun_out/r06/xattn_skip_padded_ab.log
