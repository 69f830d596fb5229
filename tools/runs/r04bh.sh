#!/bin/bash
# the nccl (RCCL) branch of the bench on one rank, inference and training, with the round-4 defaults
O=gpurun_out/r04bh; mkdir -p $O
MRX_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 4 --warmup 1 > $O/dist_inf.json 2> $O/dist_inf.err; tail -c 600 $O/dist_inf.json | head -c 400; echo
MRX_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29518 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --train --dtype bf16 --steps 3 --warmup 1 > $O/dist_train.json 2> $O/dist_train.err; head -c 300 $O/dist_train.json; echo; tail -2 $O/dist_inf.err $O/dist_train.err | cut -c1-200
