#!/bin/bash
# round 6, GPU call R: occupancy of the one-term k_uconv_h (5 / 4 workgroups per CU at 1 / 2 output blocks) and 8-row items at 5 per CU against 16-row items at 3 per CU
O=gpurun_out/r06r; mkdir -p $O
R=$GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_unet_p16.py -x -q 2>&1 | tail -3 | tee $O/tests.txt
: > $O/ab.txt
for v in lib lib_v_p16th8 lib_v_p16occ0 lib lib_v_p16th8 lib_v_p16occ0; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 bench.py --model e2evn --precision 16 --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', round(r['value'],1), r['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
