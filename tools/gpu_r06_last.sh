#!/bin/bash
# round 6, closing check at the final commit: smoke(), the GEMM / attention / mx8 kernel tests on the rebuilt library, the driver's own bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python -m pytest tests/test_kernels_gpu.py tests/test_mx8_gpu.py tests/test_codeobj.py -x -q 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_fp16_driver_cmd.json 2> $O/bench_fp16_driver_cmd.err
python -c "import json;d=json.load(open('$O/bench_fp16_driver_cmd.json'));print(d['value'], d['value_dedup'], d['ms_per_step'], {k:(round(v['achieved'],1),round(v['frac'],4)) for k,v in d['roofline']['families'].items()}, d['roofline']['end_to_end_frac'], d['cpu_baseline']['value'], d.get('cpu_baseline_config0',{}).get('wall_seconds_one_stage'))"
