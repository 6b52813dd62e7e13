#!/bin/bash
# round 3: long soak / stress of the final code
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s29
mkdir -p $OUT
cd $R
timeout -k 10 500 python tests/gpu_soak.py 12000 2>&1 | grep -v amdgpu.ids | tail -9 | tee $OUT/soak_long.txt
timeout -k 10 400 python tests/gpu_mbox_stress.py 20000 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/mbox_stress_long.txt
timeout -k 10 300 python tests/gpu_build_stress.py 450 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/build_stress_long.txt
