# PMC counters of the derivative kernel (own run, kernel-trace only; no other trace domains)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -- python3 $GRAFT_REPO_ROOT/tests/gpu_kernel_bench.py pmc1 > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/tests/gpu_kernel_bench.py pmc2 > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/tcc1 -- python3 $GRAFT_REPO_ROOT/tests/gpu_kernel_bench.py pmc3 > $OUT/tcc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc2 -- python3 $GRAFT_REPO_ROOT/tests/gpu_kernel_bench.py pmc4 > $OUT/tcc2.log 2>&1
ls -R $OUT | head -40
