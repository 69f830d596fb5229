#!/bin/bash
# round 4, GPU call C: which source's no-packed-fp32 build breaks the fp32 training gradients (bisect builds), the new bf16-storage kernels and tape
O=gpurun_out/r04c; mkdir -p $O
for v in "" conv_bwd rim_layer rim_layer_wino conv conv_bf16; do
  if [ -n "$v" ]; then export MRIDC_AMD_LIB=$PWD/mridc_amd/lib_pk_$v/libmridc_amd.so; else unset MRIDC_AMD_LIB; fi
  echo "=== packed-fp32 allowed in: ${v:-nothing}" >> $O/bisect.txt
  timeout 300 python tools/probe/train_parity.py 4 48 40 f32 --quick >> $O/bisect.txt 2>&1
done
unset MRIDC_AMD_LIB
grep -E "===|whole|layers.0.convs.conv_layer.weight" $O/bisect.txt
timeout 900 python -m pytest tests/test_gpu_train_bf16.py -q -s > $O/pytest_train_bf16.txt 2>&1
tail -40 $O/pytest_train_bf16.txt
timeout 900 python tools/probe/train_parity.py > $O/train_parity_tl.txt 2>&1
tail -30 $O/train_parity_tl.txt
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 3 --warmup 1 > $O/bench_train.json 2> $O/bench_train.err
cat $O/bench_train.json | head -c 600
