#!/bin/bash
# round 4, GPU call N: the 4-wave / two-workgroups-per-CU form of the dominant layer (lib_w4: -DMRX_L2_W4) against the product library
O=gpurun_out/r04n; mkdir -p $O
W4=$PWD/mridc_amd/lib_w4/libmridc_amd.so
MRIDC_AMD_LIB=$W4 timeout 900 python -m pytest tests/test_gpu_cb8.py tests/test_gpu_headline.py tests/test_gpu_concurrent_streams.py -q -k "not training" > $O/pytest_w4.txt 2>&1; tail -4 $O/pytest_w4.txt | cut -c1-200
for i in 1 2; do
  for v in base w4; do
    if [ $v = w4 ]; then export MRIDC_AMD_LIB=$W4; else unset MRIDC_AMD_LIB; fi
    timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 20 --warmup 3 > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
    python -c "
import json,sys
d=json.loads(open('$O/bench_${v}_$i.json').read().strip().splitlines()[-1]); print('$v', round(d['value'],2), d['breakdown_ms'])"
  done
done
unset MRIDC_AMD_LIB
timeout 600 python tools/probe/side_stream_ab.py 0 > $O/side_stream_0.txt 2>&1; cat $O/side_stream_0.txt | tail -4
timeout 600 python tools/probe/side_stream_ab.py 14 > $O/side_stream_14.txt 2>&1; cat $O/side_stream_14.txt | tail -4
