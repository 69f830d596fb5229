#!/bin/bash
# round 6, GPU call D: (1) precision-16 layer 2 with h_prev requested at chunk pair 0 / 1 / 2 / 3 (A/B builds; streaming policy on);  (2) the fp32-class layer
# kernels with the streaming (nt) cache policy on their state streams: all four streams, all but layer 1's stores, loads only -- per kernel and as the headline line
O=gpurun_out/r06d; mkdir -p $O
R=$GRAFT_REPO_ROOT
: > $O/time.txt
for v in lib lib_v_hp0 lib_v_hp2 lib_v_hp3 lib; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so PROBE_FP32=0 python3 tools/probe/amp16_time.py >> $O/time.txt 2>&1
done
for v in lib lib_v_f32nt lib_v_f32ntk lib_v_f32ntl lib; do
  [ -f $R/mridc_amd/$v/libmridc_amd.so ] && MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 tools/probe/l2_time.py >> $O/time.txt 2>&1
done
grep -v amdgpu.ids $O/time.txt
for v in lib lib_v_f32nt lib_v_f32ntk lib_v_f32ntl lib; do
  [ -f $R/mridc_amd/$v/libmridc_amd.so ] && MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['value'], r['breakdown_ms'])" >> $O/headline.txt
done
cat $O/headline.txt
python3 bench.py --precision 16 --no-other-configs --steps 10 --warmup 3 --cpu-slices 1 --cpu-cascades 1 > $O/bench_p16.json 2> $O/bench_p16.err
python3 -c "import json; r=json.load(open('$O/bench_p16.json')); print(r['value'], r['breakdown_ms'], r.get('cpu_baseline'), r.get('parity_vs_oracle'))"
