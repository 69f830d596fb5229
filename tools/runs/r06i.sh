#!/bin/bash
# round 6, GPU call I: precision-16 line -- layer 1's stores with the default cache policy (layer 2 reads them right back) against streaming, at 8 / 4 slices per launch
O=gpurun_out/r06i; mkdir -p $O
R=$GRAFT_REPO_ROOT
: > $O/ab.txt
for cfg in "--batch 8 --streams 2" "--batch 4 --streams 2" "--batch 4 --streams 4" "--batch 8 --streams 3"; do
for v in lib lib_v_l1st0 lib lib_v_l1st0; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 bench.py --precision 16 --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 $cfg 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v $cfg', round(r['value'],2), r['breakdown_ms'])" >> $O/ab.txt
done
done
cat $O/ab.txt
