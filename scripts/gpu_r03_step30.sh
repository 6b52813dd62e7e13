#!/bin/bash
# round 3: per-wave LDS staging of the neighbour records (prototype, -DNDT_LDS_STAGE) against production
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s30
mkdir -p $OUT
cd $R
rm -rf $R/slam-sam_amd/csrc/build-ab $R/slam-sam_amd/libndt_hip_ab.so
make -C $R/slam-sam_amd/csrc VARIANT=ab EXTRA="-DNDT_LDS_STAGE" -j8 > $OUT/make.log 2>&1 || { tail -5 $OUT/make.log; exit 1; }
for rep in 1 2; do
  timeout -k 10 120 python tests/gpu_abl_bench.py "production" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lds.txt
  NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_ab.so timeout -k 10 120 python tests/gpu_abl_bench.py "-DNDT_LDS_STAGE" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lds.txt
done
for rep in 1 2; do
  timeout -k 10 120 python tests/gpu_step_ab.py "production" 2>&1 | grep -v amdgpu.ids | cut -c1-215 | tee -a $OUT/lds.txt
  NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_ab.so timeout -k 10 120 python tests/gpu_step_ab.py "-DNDT_LDS_STAGE" 2>&1 | grep -v amdgpu.ids | cut -c1-215 | tee -a $OUT/lds.txt
done
rm -rf $R/slam-sam_amd/csrc/build-ab $R/slam-sam_amd/libndt_hip_ab.so
