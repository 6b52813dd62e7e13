#!/bin/bash
# round 3: XCD-aware chunk assignment in k_derivatives -- interleaved A/B (kernel time by dispatch events, step wall time)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s22
mkdir -p $OUT
cd $R
for rep in 1 2 3; do
  timeout -k 10 120 python tests/gpu_abl_bench.py "XCD-aware chunks" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/xcd.txt
  NDT_DERIV_XCD=0 timeout -k 10 120 python tests/gpu_abl_bench.py "chunk = block id" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/xcd.txt
done
for rep in 1 2 3; do
  timeout -k 10 120 python tests/gpu_step_ab.py "XCD-aware chunks" 2>&1 | grep -v amdgpu.ids | cut -c1-215 | tee -a $OUT/xcd.txt
  NDT_DERIV_XCD=0 timeout -k 10 120 python tests/gpu_step_ab.py "chunk = block id" 2>&1 | grep -v amdgpu.ids | cut -c1-215 | tee -a $OUT/xcd.txt
done
