#!/bin/bash
# Round-3 evidence, everything under gpurun_out/art3/: bench lines (N = 1; 2 ranks on the one device; forced
# multi-rank path with one rank incl. RCCL), rocprofv3 kernel stats of the bench command, PMC counters (separate
# passes), size sweep, modes, upload, replay.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/art3
rm -rf $OUT; mkdir -p $OUT
cd $R
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/host.txt
timeout -k 10 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
NDT_BENCH_SINGLE_DEVICE=1 NDT_BENCH_PROBE=0 timeout -k 10 400 python bench.py --gpus 2 --no-cpu-baseline > $OUT/bench_2on1.json 2> $OUT/bench_2on1.err; echo "bench 2on1 rc=$?"
NDT_BENCH_FORCE_DIST=1 NDT_BENCH_PROBE=0 timeout -k 10 400 python bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "bench dist1 rc=$?"
timeout -k 10 300 python tests/gpu_size_sweep.py both 2>&1 | grep -v amdgpu.ids > $OUT/size_sweep.txt
timeout -k 10 300 python tests/gpu_modes_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/modes.txt
timeout -k 10 200 python tests/gpu_upload_bench.py 2>&1 | tail -4 > $OUT/upload.txt
timeout -k 10 300 python -m pytest tests/test_gpu_replay.py -m gpu -q -s 2>&1 | grep "C5" > $OUT/replay.txt
cd /tmp && export TMPDIR=/tmp
export NDT_BENCH_PROBE=0 NDT_BENCH_HOST_CLOUD=0   # profile the headline workload only
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof.log 2>&1
unset NDT_BENCH_PROBE NDT_BENCH_HOST_CLOUD
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
P=$OUT/pmc; mkdir -p $P
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $P/sq1 -- python3 $R/tests/gpu_kernel_bench.py pmc1 > $P/sq1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $P/sq2 -- python3 $R/tests/gpu_kernel_bench.py pmc2 > $P/sq2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/tcc1 -- python3 $R/tests/gpu_kernel_bench.py pmc3 > $P/tcc1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/tcc2 -- python3 $R/tests/gpu_kernel_bench.py pmc4 > $P/tcc2.log 2>&1
python3 $R/scripts/pmc_summary.py $P > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/prof/*/*trace* 2>/dev/null
find $P -name "*kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
cd $R
timeout -k 10 300 python tests/gpu_soak.py 2>&1 | grep -v amdgpu.ids | tail -8 > $OUT/soak.txt
timeout -k 10 200 python tests/gpu_mbox_stress.py 3000 2>&1 | grep -v amdgpu.ids | tail -3 > $OUT/mbox_stress.txt
timeout -k 10 300 python tests/gpu_build_stress.py 150 2>&1 | grep -v amdgpu.ids | tail -3 > $OUT/build_stress.txt
timeout -k 10 120 python tests/gpu_abl_bench.py "f64 records" 2>&1 | grep -v amdgpu.ids > $OUT/packed.txt
NDT_ABL_PACKED=1 timeout -k 10 120 python tests/gpu_abl_bench.py "packed 48-byte records" 2>&1 | grep -v amdgpu.ids >> $OUT/packed.txt
make -C $R/slam-sam_amd/csrc VARIANT=stamps -j8 > /dev/null 2>&1
NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_stamps.so timeout -k 10 200 python tests/gpu_stamps_prelaunch.py 2>&1 | grep -v amdgpu.ids | tail -9 > $OUT/stamps_prelaunch.txt
NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_stamps.so timeout -k 10 200 python tests/gpu_build_stamps.py 2>&1 | grep -v amdgpu.ids | tail -18 > $OUT/build_stamps.txt
rm -f $R/slam-sam_amd/libndt_hip_stamps.so
echo done; head -c 700 $OUT/bench.json; echo; cat $OUT/replay.txt; cat $OUT/modes.txt | tail -8
