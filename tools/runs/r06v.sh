#!/bin/bash
# round 6, GPU call V (lib 266): mrx_conv1x1_sq_p16 and mrx_conv_sbs_p16 (the qRIM's 128 -> 128 cells and its 5x5 first layer on one fp16 term): tests, qCIRIM throughput in both precisions
O=gpurun_out/r06v; mkdir -p $O
python3 -m pytest tests/test_gpu_unet_p16.py tests/test_gpu_conv.py tests/test_gpu_headline.py -x -q -k "precision16 or qcirim or 1x1 or cell or sbs" 2>&1 | tail -12 | tee $O/tests.txt
: > $O/ab.txt
for v in 32 16 32 16; do
  python3 bench.py --model qcirim --streams 4 --precision $v --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('precision $v', round(r['value'],1), r['ms_per_step'])" >> $O/ab.txt
done
python3 bench.py --model qcirim --streams 4 --precision 16 --no-other-configs --steps 10 --warmup 2 --cpu-slices 1 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['parity_vs_oracle'], r['roofline'].get('avg_ms'))" >> $O/ab.txt
cat $O/ab.txt
