#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r02_t5.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_t5.log
python tests/gpu_r02_ab.py c3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_ab3.txt
for v in "wide" "wide shuffle"; do python tests/gpu_wide_bench.py $v 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02_wide_sorted.txt
