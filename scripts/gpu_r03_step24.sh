#!/bin/bash
# round 3: XCD-aware chunks -- full suite, then the size sweep with and without
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s24
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/full.log 2>&1; rc=$?; tail -3 $OUT/full.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tests/gpu_size_sweep.py both 2>&1 | grep -v amdgpu.ids > $OUT/sweep_xcd.txt
NDT_DERIV_XCD=0 timeout -k 10 300 python tests/gpu_size_sweep.py both 2>&1 | grep -v amdgpu.ids > $OUT/sweep_plain.txt
paste -d'|' <(cut -c1-62 $OUT/sweep_xcd.txt) <(cut -c30-62 $OUT/sweep_plain.txt)
