#!/bin/bash
# round 6, GPU call G: the workgroup-tile form of the precision-16 first layer against the wave-private form (A/B builds + phase ablation), its tests, then call F (general-mask passes)
O=gpurun_out/r06g; mkdir -p $O
R=$GRAFT_REPO_ROOT
: > $O/time.txt
for v in lib lib_v_l1wave lib lib_v_l1wave; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so PROBE_FP32=0 python3 tools/probe/amp16_time.py >> $O/time.txt 2>&1
done
MRIDC_AMD_LIB=$R/mridc_amd/lib_probe/libmridc_amd.so PROBE_FP32=0 PROBE_ABL=1,2,4,3,6,7 python3 tools/probe/amp16_time.py >> $O/time.txt 2>&1
grep -v amdgpu.ids $O/time.txt
python3 -m pytest tests/test_gpu_amp16.py -x -q -s 2>&1 | tail -14 | tee $O/amp16_tests.txt
python3 bench.py --precision 16 --no-other-configs --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('precision16', r['value'], r['breakdown_ms'])" | tee $O/p16.txt
bash tools/runs/r06f.sh
