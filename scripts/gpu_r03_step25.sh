#!/bin/bash
# round 3: block shape for the per-rank source sizes of a multi-GPU job (25 k .. 100 k points)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s25
mkdir -p $OUT
cd $R
for b in 0 128 192 256 384; do
  echo "== NDT_DERIV_BLOCK=$b (0 = default)" | tee -a $OUT/small_blocks.txt
  NDT_DERIV_BLOCK=$b timeout -k 10 200 python tests/gpu_size_sweep.py 2>&1 | grep -v amdgpu.ids | grep -E "n= +(12500|25000|50000|100000) " | tee -a $OUT/small_blocks.txt
done
