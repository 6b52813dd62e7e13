#!/bin/bash
# fused sort passes: parity, A/B against the classic passes, kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_keyframes.py tests/test_gpu_features.py -m gpu -q -x > gpurun_out/r02_t6.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02_t6.log
[ $rc -eq 0 ] || exit $rc
python tests/gpu_build_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_build_fused_ab.txt
python tests/gpu_r02_ab.py fused 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_ab3.txt
bash scripts/gpu_r02_prof.sh c
