#!/bin/bash
# interleaved A/B of what a pre-launched evaluation kernel does while it waits for its pose
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
  for cfg in "0 0" "1 0" "0 1"; do
    set -- $cfg
    NDT_MBOX_PREFETCH=$1 NDT_MBOX_PRELOAD=$2 python tests/gpu_r02_ab.py "prefetch=$1,preload=$2" 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r02_mailbox_prefetch_ab.txt
