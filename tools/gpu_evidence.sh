#!/bin/bash
# Evidence at HEAD, one GPU call (the LAST call of a round):  gpurun --timeout 3000 -- 'bash tools/gpu_evidence.sh'
# rocprofv3 kernel statistics of one eager bench step, the PMC traffic of the kernel bench.py's roofline names, the GEMM / MX-fp8 GEMM / attention PMC
# summaries, the attention microbenchmark, the library comparison, hot-vs-cold operands, the packing-time probe, the fp8 line.
# Everything lands under gpurun_out/${ROUND:-r06}/ (copy to profiles/ to commit).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${ROUND:-r06}
mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph --no-power --dedup-steps 0 > /tmp/rp_stats.json 2> /tmp/rp_stats.err )
f=$(find /tmp/rp_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprofv3_kernel_stats_fp16.csv
head -c 300 /tmp/rp_stats.json; echo; head -14 $O/rocprofv3_kernel_stats_fp16.csv | cut -c1-150
python tools/pmc_traffic.py fp16 $O/pmc_traffic_fp16.json 2>&1 | tail -2
python tools/pmc_gemm.py $O/pmc_gemm.json 2>&1 | tail -2
python tools/pmc_gemm.py mx8 $O/pmc_mx8.json 2>&1 | tail -2
python tools/pmc_attention.py $O/pmc_attention.json 2>&1 | tail -2
timeout 300 python tools/attn_bench.py 2>&1 | grep -v libdrm | tee $O/attn_bench.log | tail -8
timeout 400 python tools/vs_hipblaslt.py --rounds 3 2>&1 | grep -v libdrm | tee $O/vs_hipblaslt.log | tail -11
timeout 400 python tools/gemm_cold.py --rounds 3 2>&1 | grep -v libdrm | tee $O/gemm_hot_cold.log | tail -7
timeout 300 python tools/pack_time.py 2>&1 | grep -v libdrm | tee $O/pack_time.log | tail -5
python bench.py --gpus 1 --steps 2 --warmup 1 --dtype fp8 --no-cpu-baseline --by-shape $O/by_shape_fp8_v1.txt > $O/bench_fp8_v1.json 2> $O/bench_fp8_v1.err
head -c 300 $O/bench_fp8_v1.json; echo
