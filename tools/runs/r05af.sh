#!/bin/bash
# round 5, GPU call AF: ConvTranspose2d 2x2 on the matrix pipe (k_uconvT_h) -- U-Net / model tests, the kernel's own duration, the E2EVN line alternating
O=gpurun_out/r05af; mkdir -p $O
R=$GRAFT_REPO_ROOT
MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_uct_h/libmridc_amd.so timeout 600 python -m pytest tests/test_gpu_unet_fused.py tests/test_gpu_models.py tests/test_n4_models.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2; do
  for v in uct_cur uct_h; do
    MRIDC_AMD_LIB=$PWD/mridc_amd/lib_v_$v/libmridc_amd.so timeout 600 python bench.py --model e2evn --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$v', round(r['value'], 1), r['ms_per_step'], (r.get('parity_vs_oracle') or {}).get('rel_l2'))" | tee -a $O/e2evn.txt
  done
done
cd /tmp && export TMPDIR=/tmp
export MRIDC_AMD_LIB=$R/mridc_amd/lib_v_uct_h/libmridc_amd.so
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --model e2evn --steps 4 --warmup 1 --graph 0 --streams 1 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/uct_h_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
grep "k_uconvT" $R/$O/uct_h_kernel_stats.md | cut -c1-130
