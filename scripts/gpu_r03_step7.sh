#!/bin/bash
# round 3: record locality after the bucketed build (leaf slots no longer in cell order): 80-byte vs 128-byte records, vs the sort-based build
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
for rep in 1 2; do
  timeout -k 5 120 python tests/gpu_step_ab.py "bucketed, 80 B records" 2>&1 | grep -v amdgpu.ids
  NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_rec128.so timeout -k 5 120 python tests/gpu_step_ab.py "bucketed, 128 B records" 2>&1 | grep -v amdgpu.ids
  NDT_BUCKET_BUILD=0 timeout -k 5 120 python tests/gpu_step_ab.py "sort-based (cell order)" 2>&1 | grep -v amdgpu.ids
  NDT_BUCKET_BUILD=0 NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_rec128.so timeout -k 5 120 python tests/gpu_step_ab.py "sort-based, 128 B" 2>&1 | grep -v amdgpu.ids
done | tee $OUT/step_ab3.txt
