#!/bin/bash
# bench rehearsals: the default line, the forced-distributed path with one rank (process group +
# shm + RCCL legs), and two ranks sharing the one device through the shared-memory reducer
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r02_bench.json
NDT_BENCH_FORCE_DIST=1 NDT_BENCH_PROBE=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02_bench_dist1.json 2> gpurun_out/r02_bench_dist1.err; echo "forced-dist rc=$?"; tail -c 1500 gpurun_out/r02_bench_dist1.json; tail -5 gpurun_out/r02_bench_dist1.err
NDT_BENCH_SINGLE_DEVICE=1 NDT_BENCH_PROBE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r02_bench_2on1.json 2> gpurun_out/r02_bench_2on1.err; echo "2-on-1 rc=$?"; tail -c 1200 gpurun_out/r02_bench_2on1.json
