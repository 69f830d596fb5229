#!/bin/bash
# round 6, GPU call M: the workgroup-tile form of the fp32-class first layer (k_rim_layer1_t) against the sixteen-wave form: bit-identity tests, isolated kernel, headline
O=gpurun_out/r06m; mkdir -p $O
R=$GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_cb8.py tests/test_gpu_headline.py tests/test_gpu_conv.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
: > $O/ab.txt
for v in lib lib_v_l1wave lib lib_v_l1wave; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 tools/probe/l2_time.py 2>&1 | grep -v amdgpu >> $O/ab.txt
done
for v in lib lib_v_l1wave lib lib_v_l1wave; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', round(r['value'],2), r['breakdown_ms'])" >> $O/ab.txt
done
cat $O/ab.txt
