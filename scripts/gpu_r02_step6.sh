#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02_t8.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r02_t8.log
for i in 1 2; do
  NDT_PRELAUNCH=0 timeout -k 10 120 python tests/gpu_r02_ab.py classic 2>&1 | grep -v amdgpu.ids
  NDT_PRELAUNCH=1 timeout -k 10 120 python tests/gpu_r02_ab.py prelaunch 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02_ab_prelaunch.txt
