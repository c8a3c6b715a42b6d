#!/bin/bash
# the whole -m gpu suite at HEAD, log only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r06/pytest_gpu_head.log
