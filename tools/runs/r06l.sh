#!/bin/bash
# round 6, GPU call L: the precision-16 line with the streaming policy off / loads only / stores only / both (states are updated IN PLACE in the loop; the isolated
# kernel timings that chose the policy wrote to a second buffer, and a pure in-place copy is slower with nt: tools/probe/state_stream_probe.hip)
O=gpurun_out/r06l; mkdir -p $O
R=$GRAFT_REPO_ROOT
: > $O/ab.txt
for rep in 1 2; do
for v in lib lib_v_ampnont lib_v_ampntld lib_v_ampntst; do
  MRIDC_AMD_LIB=$R/mridc_amd/$v/libmridc_amd.so python3 bench.py --precision 16 --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', round(r['value'],2), r['breakdown_ms'])" >> $O/ab.txt
done
done
cat $O/ab.txt
