#!/bin/bash
# round 6, GPU call F: the three passes of the general-mask gradient -- durations at 1 / 2 / 4 / 8 slices per launch (does the exchange leaving the 256 MB memory-side
# cache change the time per slice?) and the SQ wave-state counters per pass (parked on memory, or issuing vector instructions?)
O=gpurun_out/r06f; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/$O/times.txt
for b in 1 2 4 8; do
  PROBE_TIME=1 python3 $R/tools/probe/llg2d_passes.py $b 2>/dev/null | grep "^B" >> $R/$O/times.txt
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/tr$b -o t -- python3 $R/tools/probe/llg2d_passes.py $b > $R/$O/tr$b.log 2>&1
  python3 $R/tools/rocpd_summary.py $R/$O/tr$b/t_results.db 2>/dev/null | grep "k_pfa372_expand\|k_cols_dc\|k_pfa372_reduce" | cut -c1-60,150-260 | sed "s/^/B $b /" >> $R/$O/times.txt
  rm -rf $R/$O/tr$b
done
cat $R/$O/times.txt
ARGS=""
for b in 1 4 8; do
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES -d $R/$O/pmc$b -o p --output-format csv -- python3 $R/tools/probe/llg2d_passes.py $b > $R/$O/pmc$b.log 2>&1
  f=$(ls $R/$O/pmc$b/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && ARGS="$ARGS B$b=$f"
done
python3 $R/tools/probe/llg2d_passes_summary.py $ARGS > $R/$O/pass_counters.md 2> $R/$O/summary.err
cat $R/$O/pass_counters.md; tail -3 $R/$O/pmc4.log
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
