#!/bin/bash
# round 3: dedicated summing block, 3- vs 4-deep record pipeline, block shapes -- interleaved A/B on one box
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py tests/test_gpu_trajectory.py tests/test_gpu_multiproc.py tests/test_svn.py -m gpu -x -q > $OUT/t6.log 2>&1; rc=$?
tail -6 $OUT/t6.log
[ $rc -ne 0 ] && exit $rc
D3=$R/slam-sam_amd/libndt_hip_depth3.so
for rep in 1 2; do
  timeout -k 5 120 python tests/gpu_step_ab.py "dedicated summer" 2>&1 | grep -v amdgpu.ids
  NDT_DERIV_DEDICATED=0 timeout -k 5 120 python tests/gpu_step_ab.py "block 0 sums" 2>&1 | grep -v amdgpu.ids
  NDT_HIP_LIB=$D3 timeout -k 5 120 python tests/gpu_step_ab.py "dedicated, depth 3" 2>&1 | grep -v amdgpu.ids
  NDT_DERIV_BLOCK=1024 timeout -k 5 120 python tests/gpu_step_ab.py "dedicated, 1024" 2>&1 | grep -v amdgpu.ids
  NDT_DERIV_BLOCK=1024 NDT_HIP_LIB=$D3 timeout -k 5 120 python tests/gpu_step_ab.py "ded, 1024, depth 3" 2>&1 | grep -v amdgpu.ids
done | tee $OUT/step_ab2.txt
