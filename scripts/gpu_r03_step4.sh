#!/bin/bash
# round 3: pre-launched kernels on alternating streams -- correctness first, then interleaved A/B
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_features.py tests/test_gpu_trajectory.py tests/test_gpu_multiproc.py -m gpu -x -q > $OUT/t4.log 2>&1; rc=$?
tail -8 $OUT/t4.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  NDT_PRELAUNCH_STREAMS=2 timeout -k 5 120 python tests/gpu_step_ab.py "two streams" 2>&1 | grep -v amdgpu.ids
  NDT_PRELAUNCH_STREAMS=1 timeout -k 5 120 python tests/gpu_step_ab.py "one stream" 2>&1 | grep -v amdgpu.ids
  NDT_PRELAUNCH=0 timeout -k 5 120 python tests/gpu_step_ab.py "no pre-launch" 2>&1 | grep -v amdgpu.ids
done | tee $OUT/prelaunch_streams_ab.txt
timeout -k 5 200 python tests/gpu_mbox_stress.py 3000 2>&1 | grep -v amdgpu.ids | tail -5 | tee $OUT/mbox_stress.txt
