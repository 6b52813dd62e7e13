#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r02_t7.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_t7.log
for i in 1 2; do python tests/gpu_r02_ab.py waveflow_tpl 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02_ab_waveflow2.txt
bash scripts/gpu_r02_prof.sh wf2 | head -3
bash scripts/gpu_stamps.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_stamps_waveflow.txt
