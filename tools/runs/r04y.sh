#!/bin/bash
O=gpurun_out/r04y; mkdir -p $O; R=$GRAFT_REPO_ROOT
for g in 1 0 1; do timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 4 --warmup 1 --train-graph $g > $O/bench_train_g$g.json 2> $O/bench_train_g$g.err; head -c 230 $O/bench_train_g$g.json; echo; tail -2 $O/bench_train_g$g.err | cut -c1-300; done
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype f32 --steps 3 --warmup 1 --train-graph 1 > $O/bench_train_f32_g1.json 2> $O/bench_train_f32_g1.err; head -c 230 $O/bench_train_f32_g1.json; echo; tail -2 $O/bench_train_f32_g1.err | cut -c1-300
