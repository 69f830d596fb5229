#!/bin/bash
# round 6, GPU call U: k_conv1x1_sb128 with a wave's first segment requested before the weight staging and h_prev under the matrix work (lib) against the serial form (lib_v_c1old)
O=gpurun_out/r06u; mkdir -p $O
R=$GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_headline.py -x -q -k "1x1 or qcirim or gated or sb128 or cell" 2>&1 | tail -4 | tee $O/tests.txt
: > $O/ab.txt
for v in lib lib_v_c1old lib lib_v_c1old; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 bench.py --model qcirim --streams 4 --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', round(r['value'],1), r['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
