#!/bin/bash
# build A/B of the ticket-bearing launch shapes + parity subset with all knobs on
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tests/gpu_build_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_build_tickets.txt
NDT_BOUNDS_BLOCKS=256 NDT_BOUNDS_UNROLL=8 NDT_RUN_KEYS=16 NDT_FINALIZE_THREADS=256 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_keyframes.py -m gpu -q -x > gpurun_out/r02_t5.log 2>&1; echo "pytest (all knobs) rc=$?"; tail -3 gpurun_out/r02_t5.log
