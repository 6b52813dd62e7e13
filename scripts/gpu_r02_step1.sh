#!/bin/bash
# round 2, first perf checkpoint: parity suite, then A/B of the final-sum variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r02_t3.log 2>&1; echo "pytest rc=$?" ; tail -5 gpurun_out/r02_t3.log
for i in 1 2; do
  NDT_DERIV_SUMMER=0 python tests/gpu_r02_ab.py ticket 2>&1 | grep -v amdgpu.ids
  NDT_DERIV_SUMMER=1 python tests/gpu_r02_ab.py summer 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02_ab1.txt
