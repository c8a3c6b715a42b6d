#!/bin/bash
# round 6, first GPU call:  gpurun --timeout 3000 -- 'bash tools/gpu_r06_first.sh'
# BASELINE configs[0] end to end at full width against the fp32 oracle loop and its fp16-emulated twin, the full-width fused
# step at 512^2 (driver-run from now on) and at 1024^2 with the emulated twin (missing 6).  ~30 minutes, almost all of it host-oracle time.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -m pytest "tests/test_fullsize_properties_gpu.py::test_one_fused_step_at_full_width_matches_the_oracle[512]" -x -q -s --durations=3 > $O/fused_step_512.log 2>&1
tail -5 $O/fused_step_512.log
OMG_RUN_SLOW=1 python -m pytest tests/test_config0_fullwidth_gpu.py -x -q -s --durations=3 > $O/config0_fullwidth.log 2>&1
tail -5 $O/config0_fullwidth.log
OMG_RUN_SLOW=1 python -m pytest "tests/test_fullsize_properties_gpu.py::test_one_fused_step_at_full_width_matches_the_oracle[1024]" -x -q -s --durations=3 > $O/fused_step_1024.log 2>&1
tail -5 $O/fused_step_1024.log
