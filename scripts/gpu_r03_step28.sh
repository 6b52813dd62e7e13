#!/bin/bash
# round 3: in-kernel stamps of a pre-launched evaluation on a rank's share of the scan (25 k points)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s28
mkdir -p $OUT
cd $R
make -C $R/slam-sam_amd/csrc VARIANT=stamps -j8 > /dev/null 2>&1
NDT_STAMPS_NSRC=25000 NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_stamps.so timeout -k 10 200 python tests/gpu_stamps_prelaunch.py 2>&1 | grep -v amdgpu.ids | tail -9 | tee $OUT/stamps_25k.txt
rm -f $R/slam-sam_amd/libndt_hip_stamps.so
