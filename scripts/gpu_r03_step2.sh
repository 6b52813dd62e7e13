#!/bin/bash
# round 3: the two-launch bucketed build -- build parity tests first, then stress, timing A/B and stamps
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py tests/test_gpu_keyframes.py tests/test_gpu_multigrid.py -m gpu -x -q -k "build or crowded or one_engine or c3 or km or fused or keyframe or multigrid or tags" > $OUT/t2.log 2>&1; rc=$?
tail -25 $OUT/t2.log
[ $rc -ne 0 ] && exit $rc
timeout -k 5 300 python tests/gpu_build_stress.py 150 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/build_stress.txt
timeout -k 10 600 python tests/gpu_build_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/build_ab.txt
NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_stamps.so timeout -k 10 300 python tests/gpu_build_stamps.py 2>&1 | grep -v amdgpu.ids | tail -18 | tee $OUT/build_stamps.txt
