#!/bin/bash
# round 3: rehearsal of the driver's N > 1 launch line (torch.distributed.run, 2 ranks on the box's one device)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s18
mkdir -p $OUT
cd $R
export NDT_BENCH_SINGLE_DEVICE=1 NDT_BENCH_PROBE=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_torchrun2.json 2> $OUT/bench_torchrun2.err; echo "torchrun rc=$?"
tail -3 $OUT/bench_torchrun2.err
python -c "
import json
d=json.loads([l for l in open('$OUT/bench_torchrun2.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['reduce'], d['config']['launch'], d['config']['reduce_variants'], d.get('reduce_failed'))"
