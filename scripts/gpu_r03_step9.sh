#!/bin/bash
# round 3: rehearsal of the multi-rank paths with 2 and 4 ranks on the box's one device (shm + peer-write; RCCL refuses duplicate GPUs)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
for n in 2 4; do
  NDT_BENCH_SINGLE_DEVICE=1 NDT_BENCH_PROBE=0 timeout -k 10 400 python bench.py --gpus $n --no-cpu-baseline > $OUT/bench_${n}on1.json 2> $OUT/bench_${n}on1.err; echo "bench ${n}on1 rc=$?"
  python -c "
import json;d=json.load(open('$OUT/bench_${n}on1.json'));print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['reduce'], json.dumps(d['config']['reduce_variants']), d['final_error_vs_ground_truth'])"
done
timeout -k 10 300 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -3
