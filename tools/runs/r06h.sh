#!/bin/bash
# round 6, GPU call H: streaming cache policy on the sensitivity-map loads of mrx_llg372 (A/B builds), fp32-class headline and the precision-16 line, alternating
O=gpurun_out/r06h; mkdir -p $O
R=$GRAFT_REPO_ROOT
: > $O/ab.txt
for v in lib lib_v_llgnt lib lib_v_llgnt; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v fp32-class', round(r['value'],2), r['breakdown_ms'])" >> $O/ab.txt
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 bench.py --precision 16 --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v precision16', round(r['value'],2), r['breakdown_ms'])" >> $O/ab.txt
done
cat $O/ab.txt
