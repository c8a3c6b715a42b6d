#!/bin/bash
# Round 5, fifth GPU call: (1) self-attention in situ, graph replay: v7 (row-major V) vs v3 + transpose_v; (2) residual launches through the persistent
# generic form (debug bit 8) vs the LDS-staged one-tile-per-block form; (3) the evidence set at HEAD; (4) the slow tests (full-width fused step vs oracle)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
B="--steps 2 --warmup 1 --dedup-steps 0 --no-cpu-baseline --no-roofline"
timeout 400 python bench.py $B > $O/fifth_bench_v7.json 2> $O/fifth_bench_v7.err; head -c 220 $O/fifth_bench_v7.json; echo
OMG_ATTN_ROW_MAJOR_V=0 timeout 400 python bench.py $B > $O/fifth_bench_v3.json 2> $O/fifth_bench_v3.err; head -c 220 $O/fifth_bench_v3.json; echo
timeout 400 python bench.py $B > $O/fifth_bench_v7b.json 2> $O/fifth_bench_v7b.err; head -c 220 $O/fifth_bench_v7b.json; echo
timeout 300 python tools/ksched_ab.py 25,65561 3 slots 2>&1 | grep -v libdrm | tee $O/fifth_res_form_ab_slots.log
timeout 300 python tools/ksched_ab.py 25,65561 3 n640 2>&1 | grep -v libdrm | tee $O/fifth_res_form_ab_n640.log
OMG_GEMM_VARIANT=65561 timeout 400 python bench.py $B > $O/fifth_bench_res4.json 2> $O/fifth_bench_res4.err; head -c 220 $O/fifth_bench_res4.json; echo
ROUND=r05 bash tools/gpu_evidence.sh
OMG_RUN_SLOW=1 timeout 1500 python -m pytest tests/test_fullsize_properties_gpu.py -x -q -s -k "one_fused_step_at_full_width" 2>&1 | tail -5 | tee $O/fifth_fullsize_fused_step.log
