#!/bin/bash
# interleaved A/B of the pose hand-over of pre-launched evaluations (same box, same process shape)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    NDT_MBOX_TAGGED=$1 NDT_MBOX_PRELOAD=$2 python tests/gpu_r02_ab.py "tagged=$1,preload=$2" 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r02_mailbox_ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_features.py tests/test_gpu_trajectory.py -m gpu -q -x 2>&1 | tail -3
