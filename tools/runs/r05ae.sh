#!/bin/bash
# round 5, GPU call AE: k_uconvT's own duration (kernel trace, one stream, eager) with the weights through the scalar cache / from LDS
O=gpurun_out/r05ae; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in uct_lds uct_sgpr; do
  export MRIDC_AMD_LIB=$R/mridc_amd/lib_v_$v/libmridc_amd.so
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$v -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --model e2evn --steps 4 --warmup 1 --graph 0 --streams 1 > $R/$O/prof_$v.log 2>&1
  python3 $R/tools/rocpd_summary.py $R/$O/prof_$v/t_results.db > $R/$O/${v}_kernel_stats.md 2>/dev/null
  rm -rf $R/$O/prof_$v
  echo $v; grep "k_uconvT\|k_uconv_h<1" $R/$O/${v}_kernel_stats.md | cut -c1-120
done
