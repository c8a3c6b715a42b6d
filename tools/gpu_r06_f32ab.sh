mkdir -p gpurun_out/r06
python -m pytest tests/test_vae_gpu.py -x -q -m gpu 2>&1 | tail -3
for round in 1 2; do
for v in old 4_8 0_8 4_7 2_8 6_8 1_4; do
  if [ $v = 4_8 ]; then lib=omg_amd/csrc/libomg_hip.so; else lib=tools/alt/libomg_f32_$v.so; fi
  echo "== $v $(OMG_HIP_LIB=$PWD/$lib python tools/vae_bench.py --upcast 1 --dtype fp16 2>&1 | grep -o 'gemm_f32 [0-9.]* TFLOP in [0-9.]* ms = [0-9]* TF/s\|un-instrumented.*')"
done; done 2>&1 | tee gpurun_out/r06/vae_f32_placement_ab.log
