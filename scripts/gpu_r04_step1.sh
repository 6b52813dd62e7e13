#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_handoff.py -x -q -m gpu 2>&1 | tail -15 > $OUT/s1_handoff_test.txt; echo "handoff rc=$?"
tail -5 $OUT/s1_handoff_test.txt
timeout -k 10 200 python tools/handoff_bench.py 20 2>&1 | grep -v amdgpu.ids > $OUT/s1_handoff_bench.txt
cat $OUT/s1_handoff_bench.txt
for t in 2 4 8 12; do echo "NDT_UPLOAD_THREADS=$t"; NDT_UPLOAD_THREADS=$t timeout -k 10 200 python tools/handoff_bench.py 20 2>&1 | grep "^async PointXYZI"; done | tee $OUT/s1_threads.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $OUT/s1_pytest.txt; tail -4 $OUT/s1_pytest.txt
