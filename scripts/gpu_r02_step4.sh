#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r02_t6.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_t6.log
for i in 1 2; do
  NDT_DERIV_WAVEFLOW=0 python tests/gpu_r02_ab.py barriers 2>&1 | grep -v amdgpu.ids
  NDT_DERIV_WAVEFLOW=1 python tests/gpu_r02_ab.py waveflow 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02_ab_waveflow.txt
NDT_DERIV_WAVEFLOW=1 bash scripts/gpu_r02_prof.sh wf | head -4
