#!/bin/bash
# tools/install_profiles.sh OLD NEW: copy the profile set gpurun_out/r04_$NEW (tools/runs/r04_profiles.sh) and gpurun_out/r04ad*/pytest.txt into profiles/r04_$NEW_*,
# replacing the r04_$OLD_* files
set -e
OLD=$1; NEW=$2; PY=$3; O=gpurun_out/r04_$NEW
git rm -q --cached profiles/r04_${OLD}_* 2>/dev/null || true
for n in e2evn qcirim train_bf16 mask2d headline_one_stream; do
  cp $O/${n}_kernel_stats.md profiles/r04_${NEW}_${n}_kernel_stats.md
  sed -i "1s#.*#\# rocprofv3 --kernel-trace summary of tools/runs/r04_profiles.sh (prof $n)#" profiles/r04_${NEW}_${n}_kernel_stats.md
done
cp $O/headline_kernel_stats.md profiles/r04_${NEW}_kernel_stats.md
sed -i "1s#.*#\# rocprofv3 --kernel-trace summary of tools/runs/r04_profiles.sh (prof headline: the default line, two streams x 8 slices -- a launch's duration here includes its wait for CUs behind the other stream's persistent kernel; r04_${NEW}_headline_one_stream_kernel_stats.md is the trace the bench's HIP-event figures agree with)#" profiles/r04_${NEW}_kernel_stats.md
sed -i "1s#.*#\# rocprofv3 --kernel-trace summary of tools/runs/r04_profiles.sh (prof headline_one_stream: bench.py --streams 1, 8 slices per launch -- every kernel alone on the chip, the durations the bench's HIP events measure)#" profiles/r04_${NEW}_headline_one_stream_kernel_stats.md
sed -i "1s#.*#\# rocprofv3 --kernel-trace summary of tools/runs/r04_profiles.sh (prof e2evn: bench.py --model e2evn --steps 4 --warmup 1 --graph 0 --streams 1, 8 slices per launch)#" profiles/r04_${NEW}_e2evn_kernel_stats.md
for t in "" 8 4; do
  s=${t:+_b$t}
  python tools/pmc_summary.py $O/pmc${t}_FETCH_SIZE/*counter_collection.csv $O/pmc${t}_WRITE_SIZE/*counter_collection.csv $O/pmc${t}_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc${t}_SQ_BUSY_CU_CYCLES/*counter_collection.csv > profiles/r04_${NEW}_pmc$s.md
  cp $O/traffic$s.json profiles/r04_traffic$s.json
done
cp $PY profiles/r04_${NEW}_pytest_gpu.txt
cp $O/bench.json profiles/r04_${NEW}_bench.json
[ -f profiles/r04_${OLD}_train_bf16_one_stream_kernel_stats.md ] && mv profiles/r04_${OLD}_train_bf16_one_stream_kernel_stats.md profiles/r04_${NEW}_train_bf16_one_stream_kernel_stats.md
rm -f profiles/r04_${OLD}_*
ls profiles | grep "r04_${NEW}"
