#!/bin/bash
# round 3: smaller blocks for small sources -- full suite, the small sizes of the sweep, 2 ranks on one device
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s26
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/full.log 2>&1; rc=$?; tail -3 $OUT/full.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tests/gpu_size_sweep.py c3 2>&1 | grep -v amdgpu.ids | grep -E "n= +(1000|12500|25000|50000|100000|200000) " | tee $OUT/sweep.txt
