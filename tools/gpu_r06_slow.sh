#!/bin/bash
# round 6, late state: every slow-marked test (full-width configs[0] loop, the 1024^2 fused step with its fp16-emulated twin, the bf16 / mx8 50-step curves, the ControlNet
# mode combinations) on the kernels of HEAD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
OMG_RUN_SLOW=1 timeout 3000 python -m pytest tests -q -m "gpu and slow" --durations=12 2>&1 | tail -25 | tee $O/pytest_gpu_slow_late.log
