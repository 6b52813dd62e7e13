#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 512 640 768 832 896 1024; do
  NDT_DERIV_BLOCK=$b python tests/gpu_r02_ab.py block$b 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02_block_sweep.txt
