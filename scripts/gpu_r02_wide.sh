#!/bin/bash
# C3-wide evidence: parity test, size sweep on both maps, source-order A/B, PMC cache counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c3_wide" > gpurun_out/r02_wide_test.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_wide_test.log
python tests/gpu_size_sweep.py both 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_size_sweep.txt
for v in "c3" "c3 shuffle" "wide" "wide shuffle" "wide sorted"; do python tests/gpu_wide_bench.py $v 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02_wide_order.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_wide
rm -rf $OUT; mkdir -p $OUT
for w in c3 wide; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${w}_fetch -- python3 $GRAFT_REPO_ROOT/tests/gpu_wide_bench.py $w > $OUT/${w}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${w}_tcc -- python3 $GRAFT_REPO_ROOT/tests/gpu_wide_bench.py $w > $OUT/${w}_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/${w}_sq -- python3 $GRAFT_REPO_ROOT/tests/gpu_wide_bench.py $w > $OUT/${w}_sq.log 2>&1
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_wide_summary.txt 2>&1
tail -30 $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_wide_summary.txt
