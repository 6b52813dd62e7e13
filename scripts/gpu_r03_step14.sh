#!/bin/bash
# round 3: per-wave stamps of k_bucket_leaves' sums / statistics phases
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s14
mkdir -p $OUT
cd $R
make -C $R/slam-sam_amd/csrc VARIANT=stamps -j8 > /dev/null 2>&1
NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_stamps.so timeout -k 10 300 python tests/gpu_build_stamps.py 2>&1 | grep -v amdgpu.ids | tail -20 | tee $OUT/build_stamps.txt
rm -f $R/slam-sam_amd/libndt_hip_stamps.so
