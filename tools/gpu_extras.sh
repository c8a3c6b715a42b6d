#!/bin/bash
# Measurements beside the headline (one GPU call):   gpurun --timeout 2400 -- 'bash tools/gpu_extras.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/${ROUND:-r05}      # ROUND=r04 reproduces the names of profiles/r04_*
mkdir -p $O
: > $O/bench_extras.jsonl
python tools/bench_extras.py both --steps 1 2>&1 | grep '^{' >> $O/bench_extras.jsonl
python tools/bench_extras.py ips --ips 1,4 --steps 2 2>&1 | grep '^{' >> $O/bench_extras.jsonl
python tools/bench_extras.py config4 --dtype fp8 --steps 1 2>&1 | grep '^{' >> $O/bench_extras.jsonl
python tools/bench_extras.py config4 --steps 1 2>&1 | grep '^{' >> $O/bench_extras.jsonl
python tools/bench_extras.py instantid --steps 1 2>&1 | grep '^{' >> $O/bench_extras.jsonl
cut -c1-330 $O/bench_extras.jsonl
