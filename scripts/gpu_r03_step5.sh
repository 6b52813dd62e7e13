#!/bin/bash
# round 3: selective re-poll in the final sum -- correctness (full suite), then per-evaluation numbers and stamps
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/t5.log 2>&1; rc=$?
tail -6 $OUT/t5.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  timeout -k 5 120 python tests/gpu_step_ab.py "r03 default" 2>&1 | grep -v amdgpu.ids
  NDT_DERIV_BLOCK=1024 timeout -k 5 120 python tests/gpu_step_ab.py "block 1024" 2>&1 | grep -v amdgpu.ids
  NDT_DERIV_BLOCK=768 timeout -k 5 120 python tests/gpu_step_ab.py "block 768" 2>&1 | grep -v amdgpu.ids
done | tee $OUT/step_ab.txt
make -C slam-sam_amd/csrc VARIANT=stamps -j8 > /dev/null 2>&1
NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_stamps.so timeout -k 5 200 python tests/gpu_stamps_prelaunch.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/stamps_prelaunch.txt
timeout -k 5 300 python tests/gpu_soak.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/soak.txt
