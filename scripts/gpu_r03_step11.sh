#!/bin/bash
# round 3: four ranks on one device -- which ingredient loses rows
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
run() { name=$1; shift
  env "$@" NDT_BENCH_SINGLE_DEVICE=1 NDT_BENCH_PROBE=0 NDT_BENCH_REDUCE=shm timeout -k 10 200 python bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/tmp_4on1.json 2> $OUT/tmp_4on1.err; rc=$?
  if [ $rc -eq 0 ]; then python -c "
import json;d=json.load(open('$OUT/tmp_4on1.json'));print('%-36s' % '$name', 'ok', round(d['ms_per_step'],3), 'ms/step  build', round(d['ms_target_build'],3), 'align', round(d['ms_align'],3))"
  else echo "$name: rc=$rc $(grep -h 'NdtError:' $OUT/tmp_4on1.err | tail -1 | cut -c1-160)"; fi
}
run "no pre-launch" NDT_PRELAUNCH=0
run "no pre-launch, no dedicated" NDT_PRELAUNCH=0 NDT_DERIV_DEDICATED=0
run "no dedicated" NDT_DERIV_DEDICATED=0
run "sort-based build, classic passes" NDT_BUCKET_BUILD=0 NDT_FUSED_SORT=0
