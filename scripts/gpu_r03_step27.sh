#!/bin/bash
# round 3: the whole step on a rank's share of the scan (no exchange): new block shapes against 512-thread blocks
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s27
mkdir -p $OUT
cd $R
for n in 25000 50000 100000; do
  for rep in 1 2; do
    NDT_STEP_AB_NSRC=$n timeout -k 10 120 python tests/gpu_step_ab.py "n=$n default" 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $OUT/small_step.txt
    NDT_STEP_AB_NSRC=$n NDT_DERIV_BLOCK=512 timeout -k 10 120 python tests/gpu_step_ab.py "n=$n 512-thread blocks" 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $OUT/small_step.txt
  done
done
