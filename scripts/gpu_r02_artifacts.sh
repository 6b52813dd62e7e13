#!/bin/bash
# Round-2 evidence, everything under gpurun_out/art2/: bench line, rocprofv3 kernel stats of the
# same command, PMC counters (separate passes), size sweep on C3 and C3-wide, modes, upload,
# replay, source-order A/B on the wide map.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/art2
rm -rf $OUT; mkdir -p $OUT
cd $R
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/host.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python tests/gpu_size_sweep.py both 2>&1 | grep -v amdgpu.ids > $OUT/size_sweep.txt
python tests/gpu_modes_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/modes.txt
python tests/gpu_upload_bench.py 2>&1 | tail -4 > $OUT/upload.txt
python -m pytest tests/test_gpu_replay.py -m gpu -q -s 2>&1 | grep "C5" > $OUT/replay.txt
for v in "c3" "c3 shuffle" "wide" "wide shuffle"; do python tests/gpu_wide_bench.py $v 2>&1 | grep -v amdgpu.ids; done > $OUT/wide_order_auto.txt
cd /tmp && export TMPDIR=/tmp
export NDT_BENCH_PROBE=0 NDT_BENCH_HOST_CLOUD=0   # profile the headline workload only
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof.log 2>&1
unset NDT_BENCH_PROBE NDT_BENCH_HOST_CLOUD
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
P=$OUT/pmc; mkdir -p $P
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $P/sq1 -- python3 $R/tests/gpu_kernel_bench.py pmc1 > $P/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $P/sq2 -- python3 $R/tests/gpu_kernel_bench.py pmc2 > $P/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/tcc1 -- python3 $R/tests/gpu_kernel_bench.py pmc3 > $P/tcc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/tcc2 -- python3 $R/tests/gpu_kernel_bench.py pmc4 > $P/tcc2.log 2>&1
for w in wide; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/${w}_fetch -- python3 $R/tests/gpu_wide_bench.py $w > $P/${w}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/${w}_tcc -- python3 $R/tests/gpu_wide_bench.py $w > $P/${w}_tcc.log 2>&1
done
python3 $R/scripts/pmc_summary.py $P > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/prof/*/*trace* 2>/dev/null
find $P -name "*kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
echo done; head -c 600 $OUT/bench.json; echo; cat $OUT/replay.txt $OUT/wide_order_auto.txt
