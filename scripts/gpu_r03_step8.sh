#!/bin/bash
# round 3: KDTREE / DIRECT26 / multi-grid with 12-byte row loads of the index grid and the f32 centroid array
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/t8.log 2>&1; rc=$?
tail -6 $OUT/t8.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tests/gpu_modes_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/modes2.txt
