#!/bin/bash
# kernel-level stats of the C3 step (rocprofv3 --kernel-trace --stats), summary to gpurun_out/r02_prof_<tag>.csv
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/gpu_r02_ab.py prof_$TAG > $OUT/log.txt 2>&1
cp $(find $OUT -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r02_prof_$TAG.csv
grep -v amdgpu.ids $OUT/log.txt | tail -2
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r02_prof_$TAG.csv")))
for r in rows[:24]:
    print("%-90s calls %6s  avg %8.2f us  total %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
