#!/bin/bash
# per-kernel times of the voxel-grid build (rocprofv3 kernel trace of tests/gpu_kernel_bench.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/buildprof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/buildprof -o bp --output-format csv -- python3 $R/tests/gpu_kernel_bench.py prof > $R/gpurun_out/buildprof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/buildprof/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    n=n.split("(")[0][-60:]
    print("%-62s calls %5s avg %9.1f ns"%(n,r["Calls"],float(r["AverageNs"])))
PY
tail -1 $R/gpurun_out/buildprof.log
