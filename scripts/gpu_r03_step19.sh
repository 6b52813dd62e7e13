#!/bin/bash
# round 3: x-row load of the index grid (5 instead of 7 grid loads per point) -- A/B as a library variant, then parity
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s19
mkdir -p $OUT
cd $R
rm -rf $R/slam-sam_amd/csrc/build-ab $R/slam-sam_amd/libndt_hip_ab.so
make -C $R/slam-sam_amd/csrc VARIANT=ab EXTRA="-DNDT_XROW" -j8 > $OUT/make.log 2>&1 || { tail -5 $OUT/make.log; exit 1; }
timeout -k 10 120 python tests/gpu_abl_bench.py "production" 2>&1 | grep -v amdgpu.ids | tee $OUT/xrow.txt
NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_ab.so timeout -k 10 120 python tests/gpu_abl_bench.py "-DNDT_XROW" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/xrow.txt
timeout -k 10 120 python tests/gpu_abl_bench.py "production (again)" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/xrow.txt
NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_ab.so timeout -k 10 120 python tests/gpu_abl_bench.py "-DNDT_XROW (again)" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/xrow.txt
NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_ab.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $OUT/t.log 2>&1; tail -4 $OUT/t.log
rm -rf $R/slam-sam_amd/csrc/build-ab $R/slam-sam_amd/libndt_hip_ab.so
