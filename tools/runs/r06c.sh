#!/bin/bash
# round 6, GPU call C: the precision-16 layer kernels -- product build, phase ablations (probe build), streaming cache policy on the state streams (A/B builds),
# then the CIRIM line in that precision
O=gpurun_out/r06c; mkdir -p $O
R=$GRAFT_REPO_ROOT
python3 tools/probe/amp16_time.py > $O/time.txt 2>&1
for v in lib_v_ntst lib_v_ntboth; do
  [ -f $R/mridc_amd/$v/libmridc_amd.so ] && MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so PROBE_FP32=0 python3 tools/probe/amp16_time.py >> $O/time.txt 2>&1
done
[ -f $R/mridc_amd/lib_probe/libmridc_amd.so ] && MRIDC_AMD_LIB=$R/mridc_amd/lib_probe/libmridc_amd.so PROBE_FP32=0 PROBE_ABL=1,2,4,8,16,3,6,7,23 python3 tools/probe/amp16_time.py >> $O/time.txt 2>&1
python3 tools/probe/amp16_time.py >> $O/time.txt 2>&1
grep -v amdgpu.ids $O/time.txt
python3 bench.py --precision 16 --no-other-configs --steps 10 --warmup 3 --cpu-slices 2 > $O/bench_p16.json 2> $O/bench_p16.err
tail -c 2500 $O/bench_p16.json; tail -5 $O/bench_p16.err
