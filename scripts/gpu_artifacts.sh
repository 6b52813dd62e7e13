#!/bin/bash
# The round's evidence in ONE script (run on the GPU box through gpurun): `bash scripts/gpu_artifacts.sh NN [part ...]`
# writes everything under gpurun_out/art_rNN/; the summaries worth judging are then copied by hand into profiles/rNN_*.
# Parts (default: all): bench multi prof pmc sweeps hosts soak stamps.  Each part is bounded by its own timeout.
# (Rounds 1-3 used one throw-away driver script per experiment; they are in the git history up to c0e341e.)
N=${1:-04}; shift; PARTS=${@:-bench multi prof pmc sweeps hosts soak stamps}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/art_r$N; mkdir -p $OUT; cd $R
has() { [[ " $PARTS " == *" $1 "* ]]; }
nolog() { grep -v amdgpu.ids; }
if has bench; then   # the driver's N = 1 command
  nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/host.txt
  timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
fi
if has multi; then   # the multi-rank code path on a 1-GPU box: 2 and 4 ranks on the one device; one rank incl. RCCL
  NDT_BENCH_SINGLE_DEVICE=1 NDT_BENCH_PROBE=0 timeout -k 10 400 python bench.py --gpus 2 --no-cpu-baseline > $OUT/bench_2on1.json 2> $OUT/bench_2on1.err; echo "2on1 rc=$?"
  NDT_BENCH_SINGLE_DEVICE=1 NDT_BENCH_PROBE=0 timeout -k 10 400 python bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_4on1.json 2> $OUT/bench_4on1.err; echo "4on1 rc=$?"
  NDT_BENCH_FORCE_DIST=1 NDT_BENCH_PROBE=0 timeout -k 10 400 python bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "dist1 rc=$?"
fi
if has prof; then    # rocprofv3 kernel stats of the bench command (headline workload only)
  ( cd /tmp && export TMPDIR=/tmp NDT_BENCH_PROBE=0 NDT_BENCH_HOST_CLOUD=0 NDT_BENCH_CONFIGS=0 NDT_BENCH_PACKED=0 NDT_BENCH_PROTOCOLS=0 NDT_BENCH_CADENCE=0
    timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof.log 2>&1
    cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/prof/*/*trace* 2>/dev/null )
fi
if has pmc; then     # counters in their own passes, kernel-trace only (the pool refuses --pmc with other trace domains)
  P=$OUT/pmc; mkdir -p $P
  ( cd /tmp && export TMPDIR=/tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $P/sq1 -- python3 $R/tools/kernel_bench.py pmc1 > $P/sq1.log 2>&1
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $P/sq2 -- python3 $R/tools/kernel_bench.py pmc2 > $P/sq2.log 2>&1
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/tcc1 -- python3 $R/tools/kernel_bench.py pmc3 > $P/tcc1.log 2>&1
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/tcc2 -- python3 $R/tools/kernel_bench.py pmc4 > $P/tcc2.log 2>&1 )
  python3 scripts/pmc_summary.py $P > $OUT/pmc_summary.txt 2>&1
  find $P -name "*kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
fi
if has sweeps; then  # source-size sweep, per-mode numbers, replay
  timeout -k 10 300 python tools/size_sweep.py both 2>&1 | nolog > $OUT/size_sweep.txt
  # ... and the same sweep under a FETCH_SIZE pass per map (own runs, kernel-trace only): HBM-side bytes per launch beside each row
  for m in c3 wide; do
    ( cd /tmp && export TMPDIR=/tmp
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/sweep_pmc_$m -- python3 $R/tools/size_sweep.py $m > $OUT/sweep_pmc_$m.log 2>&1 )
    python3 tools/sweep_fetch.py $OUT/sweep_pmc_$m > $OUT/size_sweep_fetch_$m.txt 2>&1
    rm -rf $OUT/sweep_pmc_$m
  done
  timeout -k 10 300 python tools/modes_bench.py 2>&1 | nolog > $OUT/modes.txt
  timeout -k 10 300 python -m pytest tests/test_gpu_replay.py -m gpu -q -s 2>&1 | grep "C5" > $OUT/replay.txt
fi
if has hosts; then   # host hand-off: asynchronous vs blocking, the H2D probe behind the pull kernels
  timeout -k 10 200 python tools/handoff_bench.py 20 2>&1 | nolog > $OUT/handoff_bench.txt
  [ -x tools/h2d_pull_probe ] && timeout -k 5 120 ./tools/h2d_pull_probe > $OUT/h2d_pull_probe.txt 2>&1
fi
if has soak; then    # determinism soaks of the hand-offs
  timeout -k 10 300 python tools/soak.py 2>&1 | nolog | tail -8 > $OUT/soak.txt
  timeout -k 10 200 python tools/mbox_stress.py 3000 2>&1 | nolog | tail -3 > $OUT/mbox_stress.txt
  timeout -k 10 300 python tools/build_stress.py 150 2>&1 | nolog | tail -3 > $OUT/build_stress.txt
  timeout -k 10 500 python tools/handoff_stress.py 2000 2>&1 | nolog | tail -3 > $OUT/handoff_stress.txt
fi
if has stamps; then  # in-kernel 100 MHz stamps (diagnostic build, removed afterwards)
  make -C slam-sam_amd/csrc VARIANT=stamps -j8 > /dev/null 2>&1
  NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_stamps.so timeout -k 10 200 python tools/stamps_prelaunch.py 2>&1 | nolog | tail -9 > $OUT/stamps_prelaunch.txt
  NDT_HIP_LIB=$R/slam-sam_amd/libndt_hip_stamps.so timeout -k 10 200 python tools/build_stamps.py 2>&1 | nolog | tail -18 > $OUT/build_stamps.txt
  rm -f slam-sam_amd/libndt_hip_stamps.so
fi
echo done; ls $OUT
