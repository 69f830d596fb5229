#!/bin/bash
# round 5, GPU call AA: effective clock of every kernel of the headline loop, the E2EVN cascade and the training step (GRBM_GUI_ACTIVE / duration)
O=gpurun_out/r05aa; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/$O/pmc_b8 -o p --output-format csv -- python3 $R/tools/probe/pmc_r04_b8.py > $R/$O/pmc_b8.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/$O/pmc_all -o p --output-format csv -- python3 $R/tools/probe/pmc_r04.py > $R/$O/pmc_all.log 2>&1
cd $R
python tools/probe/kernel_clocks.py $O/pmc_b8 > $O/clocks_b8.txt 2>&1
python tools/probe/kernel_clocks.py $O/pmc_all > $O/clocks_all.txt 2>&1
rm -rf $O/pmc_b8 $O/pmc_all
cat $O/clocks_b8.txt | cut -c1-110; echo; head -40 $O/clocks_all.txt | cut -c1-110
