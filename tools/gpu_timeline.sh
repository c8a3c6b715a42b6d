cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
(
timeout 600 python tools/ksched_ab.py 25,16409 5 n640
timeout 600 python tools/ksched_ab.py 25,16409 3 k
echo "== plain"; timeout 120 python tools/gemm_timeline.py 65536 1280 1280 0 25
echo "== residual via LDS (EF=2)"; timeout 120 python tools/gemm_timeline.py 65536 1280 1280 0 25 res
echo "== residual register-direct (EF=0)"; timeout 120 python tools/gemm_timeline.py 65536 1280 1280 64 25 res
) 2>&1 | grep -v libdrm | tee gpurun_out/r03/ef_ab.log
