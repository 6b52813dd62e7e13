#!/bin/bash
# round 3: k_derivatives record-fetch experiments, one library variant each (HIP-event kernel time, C3, converged pose)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s15
mkdir -p $OUT
cd $R
timeout -k 10 120 python tests/gpu_abl_bench.py "production" 2>&1 | grep -v amdgpu.ids | tee $OUT/abl2.txt
for v in "-DNDT_TOUCH=1" "-DNDT_TOUCH=2" "-DNDT_PAIR_DEPTH4"; do
  rm -rf $R/slam-sam_amd/csrc/build-ab $R/slam-sam_amd/libndt_hip_ab.so
  make -C $R/slam-sam_amd/csrc VARIANT=ab EXTRA="$v" -j8 > $OUT/make.log 2>&1 || { tail -5 $OUT/make.log; exit 1; }
  NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_ab.so timeout -k 10 120 python tests/gpu_abl_bench.py "$v" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/abl2.txt
done
timeout -k 10 120 python tests/gpu_abl_bench.py "production (again)" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/abl2.txt
rm -rf $R/slam-sam_amd/csrc/build-ab $R/slam-sam_amd/libndt_hip_ab.so
