#!/bin/bash
# round 6, GPU call S: mrx_conv3x3_p16 (the qRIM's wide 3x3 convolutions on one fp16 term): tests, qCIRIM throughput in both precisions
O=gpurun_out/r06s; mkdir -p $O
python3 -m pytest tests/test_gpu_unet_p16.py tests/test_gpu_headline.py -x -q -k "precision16 or qcirim" 2>&1 | tail -12 | tee $O/tests.txt
: > $O/ab.txt
for v in 32 16 32 16; do
  python3 bench.py --model qcirim --streams 4 --precision $v --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('precision $v', round(r['value'],1), r['ms_per_step'])" >> $O/ab.txt
done
python3 bench.py --model qcirim --streams 4 --precision 16 --no-other-configs --steps 10 --warmup 2 --cpu-slices 1 > $O/qcirim16_line.json 2> $O/qcirim16.err
cp bench_detail.json $O/qcirim16_detail.json
cat $O/ab.txt; tail -c 1800 $O/qcirim16_line.json
