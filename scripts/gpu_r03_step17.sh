#!/bin/bash
# round 3: one 64-bit atomic at the end of k_bucket_leaves -- parity tests, stress, A/B, kernel stats
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s17
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py tests/test_gpu_keyframes.py -m gpu -x -q > $OUT/t.log 2>&1; rc=$?
tail -8 $OUT/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tests/gpu_build_stress.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/stress.txt
timeout -k 10 600 python tests/gpu_build_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/build_ab.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$OUT/bench.json'));print({k:d[k] for k in ('value','ms_per_step','ms_target_build','ms_align','ms_target_build_device')}, d['roofline_build']['frac'], d['roofline']['frac'])"
