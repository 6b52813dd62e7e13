#!/bin/bash
# round 3: full GPU suite with the two-launch build on, then the bench lines and the build A/B
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/t3.log 2>&1; rc=$?
tail -15 $OUT/t3.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tests/gpu_build_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/build_ab.txt
NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_stamps.so timeout -k 10 300 python tests/gpu_build_stamps.py 2>&1 | grep -v amdgpu.ids | tail -18 | tee $OUT/build_stamps.txt
timeout -k 10 400 python bench.py > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$OUT/bench2.json'));print({k:d[k] for k in ('value','ms_per_step','ms_target_build','ms_align','ms_target_build_device')}, d['roofline_build']['frac'], d['roofline']['frac'])"
