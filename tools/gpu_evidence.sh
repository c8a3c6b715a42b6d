#!/bin/bash
# Evidence at HEAD, one GPU call:  gpurun --timeout 2400 -- 'bash tools/gpu_evidence.sh'
# rocprofv3 kernel statistics of one eager bench step, the PMC traffic of the kernel bench.py's roofline names, the GEMM and attention PMC
# summaries, the attention microbenchmark, the library comparison.  Everything lands under gpurun_out/${ROUND:-r05}/ (copy to profiles/ to commit).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${ROUND:-r05}      # ROUND=r04 reproduces the names of profiles/r04_*
mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph --no-power --dedup-steps 0 > /tmp/rp_stats.json 2> /tmp/rp_stats.err )
f=$(find /tmp/rp_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprofv3_kernel_stats_fp16.csv
head -c 300 /tmp/rp_stats.json; echo; head -12 $O/rocprofv3_kernel_stats_fp16.csv | cut -c1-150
python tools/pmc_traffic.py fp16 $O/pmc_traffic_fp16.json 2>&1 | tail -2
python tools/pmc_gemm.py $O/pmc_gemm.json 2>&1 | tail -2
python tools/pmc_attention.py $O/pmc_attention.json 2>&1 | tail -2
timeout 300 python tools/attn_bench.py 2>&1 | grep -v libdrm | tee $O/attn_bench.log | tail -12
