#!/bin/bash
# round 3, first GPU step: full GPU suite (incl. the new peer-write reducer, the self-launching bench), then the bench lines
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/t1.log 2>&1; rc=$?
tail -15 $OUT/t1.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > $OUT/bench1.json 2> $OUT/bench1.err; echo "bench rc=$?"
head -c 1500 $OUT/bench1.json; echo
NDT_BENCH_SINGLE_DEVICE=1 NDT_BENCH_PROBE=0 timeout -k 10 400 python bench.py --gpus 2 --no-cpu-baseline > $OUT/bench_2on1.json 2> $OUT/bench_2on1.err; echo "bench 2on1 rc=$?"
python -c "
import json;d=json.load(open('$OUT/bench_2on1.json'));print(d['config']['reduce'], json.dumps(d['config']['reduce_variants']))"
NDT_BENCH_FORCE_DIST=1 NDT_BENCH_PROBE=0 timeout -k 10 400 python bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "bench dist1 rc=$?"
python -c "
import json;d=json.load(open('$OUT/bench_dist1.json'));print(d['config']['reduce'], json.dumps(d['config']['reduce_variants']), d['config']['rccl'])"
