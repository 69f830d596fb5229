#!/bin/bash
# round 6, GPU call T (lib 266): the GPU suite on the product build and on the CHECK build (-DMRX_CHECK_BOUNDS), smoke, the default bench line (driver-style), and the one-rank RCCL path
O=gpurun_out/r06t; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time python3 -m pytest tests -x -q -m gpu ) 2>&1 | tail -6 | tee $O/pytest_gpu.txt
( time MRIDC_AMD_LIB=$R/mridc_amd/lib_chk/libmridc_amd.so python3 -m pytest tests -x -q -m gpu ) 2>&1 | tail -6 | tee $O/pytest_gpu_check_build.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real | tee $O/bench_time.txt
wc -c $O/bench.json; python3 -c "import json; r=json.load(open('$O/bench.json')); print(json.dumps(r['summary']))"
MRX_BENCH_FORCE_DIST=1 python3 bench.py --no-other-configs --no-cpu-baseline --no-stream-inputs --steps 4 --warmup 2 2>$O/dist1.err | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('one-rank RCCL path', r['value'], r.get('world_size_seen'))" | tee $O/dist1.txt; tail -2 $O/dist1.err
