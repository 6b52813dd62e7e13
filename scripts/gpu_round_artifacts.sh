#!/bin/bash
# End-of-round evidence: bench line, rocprofv3 kernel stats of the same command, PMC counters,
# size sweep, per-mode numbers, replay.  Everything lands under gpurun_out/art/.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/art
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python tests/gpu_size_sweep.py > $OUT/size_sweep.txt 2>&1
python tests/gpu_modes_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/modes.txt
python tests/gpu_upload_bench.py 2>&1 | tail -4 > $OUT/upload.txt
python -m pytest tests/test_gpu_replay.py -m gpu -q -s 2>&1 | grep "C5" > $OUT/replay.txt
cd /tmp && export TMPDIR=/tmp
export NDT_BENCH_PROBE=0   # profile the headline workload only
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1
unset NDT_BENCH_PROBE
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
bash $R/scripts/gpu_pmc.sh > $OUT/pmc.log 2>&1
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc > $OUT/pmc_summary.txt 2>&1
echo done
