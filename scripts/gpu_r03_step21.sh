#!/bin/bash
# round 3: block shape of the 200 k-point launch once more, interleaved (default 832 x 241+1 vs 1024 x 196+1 vs 768)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s21
mkdir -p $OUT
cd $R
for rep in 1 2 3; do
  timeout -k 10 120 python tests/gpu_step_ab.py "default (832)" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/block_ab.txt
  NDT_DERIV_BLOCK=1024 timeout -k 10 120 python tests/gpu_step_ab.py "NDT_DERIV_BLOCK=1024" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/block_ab.txt
  NDT_DERIV_BLOCK=896 timeout -k 10 120 python tests/gpu_step_ab.py "NDT_DERIV_BLOCK=896" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/block_ab.txt
done
