#!/bin/bash
# round 6, last GPU call: the whole `-m gpu` suite as the driver runs it at the final commit, then the evidence set (tools/gpu_evidence.sh) and the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
( time python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest_gpu_final.log 2>&1
tail -22 $O/pytest_gpu_final.log
ROUND=r06 bash tools/gpu_evidence.sh
python bench.py --by-shape $O/by_shape_fp16_final.txt > $O/bench_fp16_final.json 2> $O/bench_fp16_final.err
head -c 400 $O/bench_fp16_final.json; echo
python -c "import json;d=json.load(open('$O/bench_fp16_final.json'));print(d['value'], d['value_dedup'], {k:(round(v['achieved'],1),round(v['frac'],4)) for k,v in d['roofline']['families'].items()}, d['roofline']['end_to_end_frac'])"
