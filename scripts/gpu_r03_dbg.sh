#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for v in "NDT_PRELAUNCH_STREAMS=2" "NDT_PRELAUNCH_STREAMS=1" "NDT_PRELAUNCH=0" "NDT_DERIV_SUMMER=0"; do
  echo "== $v"
  env $v timeout -k 5 120 python tests/gpu_step_ab.py "$v" 2>&1 | grep -v amdgpu.ids | tail -3
done
