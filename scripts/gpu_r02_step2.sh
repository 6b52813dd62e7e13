#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_keyframes.py tests/test_gpu_features.py -m gpu -q -x > gpurun_out/r02_t4.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/r02_t4.log
python tests/gpu_r02_ab.py v2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_ab2.txt
bash scripts/gpu_r02_prof.sh b
