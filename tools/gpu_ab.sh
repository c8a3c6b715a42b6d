cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout 600 python tools/ksched_ab.py $1 ${2:-5} ${4:-k} 2>&1 | grep -v libdrm | tee gpurun_out/r03/ksched_ab_$3.log
