#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_downsample.py -x -q -m gpu 2>&1 | tail -30 > $OUT/s3_pytest.txt; tail -30 $OUT/s3_pytest.txt
