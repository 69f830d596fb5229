#!/bin/bash
# round 6, GPU call P: the U-Net models' precision-16 route (mrx_unet_conv3x3_p16): parity tests, E2EVN throughput in both precisions
O=gpurun_out/r06p; mkdir -p $O
python3 -m pytest tests/test_gpu_unet_p16.py tests/test_gpu_unet_fused.py tests/test_gpu_models.py -x -q 2>&1 | tail -12 | tee $O/tests.txt
: > $O/ab.txt
for v in 32 16 32 16; do
  python3 bench.py --model e2evn --precision $v --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('precision $v', round(r['value'],1), r['ms_per_step'])" >> $O/ab.txt
done
python3 bench.py --model e2evn --precision 16 --no-other-configs --steps 6 --warmup 2 --cpu-slices 1 > $O/e2evn16_line.json 2> $O/e2evn16.err
cp bench_detail.json $O/e2evn16_detail.json
cat $O/ab.txt; tail -c 1500 $O/e2evn16_line.json
