#!/bin/bash
# round 6, fourth GPU call: the whole `-m gpu` suite as the driver runs it (no OMG_RUN_SLOW) at the state with the attention denominator landed, v3 deleted,
# the reference kwargs implemented; then the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
( time python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu_v1.log 2>&1
tail -28 $O/pytest_gpu_v1.log
python bench.py --by-shape $O/by_shape_fp16_v1.txt > $O/bench_fp16_v1.json 2> $O/bench_fp16_v1.err
head -c 700 $O/bench_fp16_v1.json; echo
python -c "import json;d=json.load(open('$O/bench_fp16_v1.json'));print(d['value'], d['value_dedup'], {k:(v['achieved'],v['frac']) for k,v in d['roofline']['families'].items()}, d['roofline']['end_to_end_frac'], d.get('cpu_baseline_config0',{}).get('wall_seconds_one_stage'))"
