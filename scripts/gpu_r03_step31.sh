#!/bin/bash
# round 3: block shape at 100 k points (a rank's share at 2 GPUs)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s31
mkdir -p $OUT
cd $R
for b in 0 448 416 640; do
  for rep in 1 2; do
    NDT_STEP_AB_NSRC=100000 NDT_DERIV_BLOCK=$b timeout -k 10 120 python tests/gpu_step_ab.py "n=100000 block=$b" 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $OUT/b100k.txt
  done
done
