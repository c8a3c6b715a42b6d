#!/bin/bash
# round 6, after the fp32-convolution / attention-descriptor / cross-attention changes: the whole -m gpu suite, then the default bench with the by-shape table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_gpu_v2.log
python bench.py --by-shape $O/by_shape_fp16_v2.txt > $O/bench_fp16_v2.json 2> $O/bench_fp16_v2.err
python -c "import json;d=json.load(open('$O/bench_fp16_v2.json'));print(d['value'], d['value_dedup'], d['ms_per_step'], {k:(round(v['achieved'],1),round(v['frac'],4)) for k,v in d['roofline']['families'].items()}, d['roofline']['end_to_end_frac'])"
