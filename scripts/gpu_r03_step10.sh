#!/bin/bash
# round 3: why two ranks on one device got slower -- knobs one at a time
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" NDT_BENCH_SINGLE_DEVICE=1 NDT_BENCH_PROBE=0 NDT_BENCH_REDUCE=shm timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline > $OUT/tmp_2on1.json 2> $OUT/tmp_2on1.err
  python -c "
import json;d=json.load(open('$OUT/tmp_2on1.json'));print('%-36s' % '$name', round(d['ms_per_step'],3), 'ms/step  build', round(d['ms_target_build'],3), 'align', round(d['ms_align'],3), 'prelaunched', d['evaluations_prelaunched_per_align'])"
}
run "default (one stream, dedicated)" X=1
run "no dedicated summer" NDT_DERIV_DEDICATED=0
run "no pre-launch" NDT_PRELAUNCH=0
run "no dedicated, no pre-launch" NDT_DERIV_DEDICATED=0 NDT_PRELAUNCH=0
run "sort-based build" NDT_BUCKET_BUILD=0
