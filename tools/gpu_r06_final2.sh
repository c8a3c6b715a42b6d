#!/bin/bash
# round 6, the LAST GPU call: evidence at HEAD (tools/gpu_evidence.sh), the tests touched since the last full suite, the VAE bench, smoke(), and the driver's own bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m pytest tests/test_vae_gpu.py tests/test_kernels_gpu.py tests/test_codeobj.py -x -q 2>&1 | tail -2 | tee $O/pytest_gpu_final2.log
python tools/vae_bench.py --upcast 1 --dtype fp16 2>&1 | grep -v libdrm | tee $O/vae_bench_upcast.log | tail -3
ROUND=r06 bash tools/gpu_evidence.sh 2>&1 | tail -60
python bench.py --gpus 1 --steps 20 --warmup 5 --by-shape $O/by_shape_fp16_final2.txt > $O/bench_fp16_driver_cmd2.json 2> $O/bench_fp16_driver_cmd2.err
python -c "import json;d=json.load(open('$O/bench_fp16_driver_cmd2.json'));print(d['value'], d['value_dedup'], d['ms_per_step'], {k:(round(v['achieved'],1),round(v['frac'],4)) for k,v in d['roofline']['families'].items()}, d['roofline']['end_to_end_frac'], d['roofline']['frac'])"
