#!/bin/bash
# round 5, GPU call W: layer 2 on operands of different toggle activity -- un-profiled times, then the effective clock per case (GRBM_GUI_ACTIVE / duration)
O=gpurun_out/r05w; mkdir -p $O
R=$GRAFT_REPO_ROOT
python tools/probe/l2_power.py > $O/l2_power.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PROBE_REPS=1 PROBE_N=20 timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/$O/pmc -o p --output-format csv -- python3 $R/tools/probe/l2_power.py > $R/$O/pmc.log 2>&1
cd $R
python tools/probe/l2_power_clock.py $O/pmc > $O/l2_power_clock.txt 2>&1
head -3 $O/pmc/*/*counter_collection.csv 2>/dev/null | cut -c1-300
rm -rf $O/pmc
cat $O/l2_power.txt | tail -5; cat $O/l2_power_clock.txt
