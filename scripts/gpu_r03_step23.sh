#!/bin/bash
# round 3: HBM fetch and L2 hit/miss of k_derivatives with and without the XCD-aware chunk assignment
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s23
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export NDT_DERIV_XCD=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$v -- python3 $R/tests/gpu_kernel_bench.py pmc3 > $OUT/f$v.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/t$v -- python3 $R/tests/gpu_kernel_bench.py pmc4 > $OUT/t$v.log 2>&1
done
unset NDT_DERIV_XCD
for v in 1 0; do echo "== NDT_DERIV_XCD=$v"; python3 $R/scripts/pmc_summary.py $OUT/f$v $OUT/t$v 2>&1 | grep -E "k_derivatives<false, 1, 1"; done | tee $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
