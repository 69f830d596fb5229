#!/bin/bash
# Round evidence in one GPU call: writes everything under gpurun_out/rNN/ (copy what is to be judged into profiles/).
#   tools/collect_profiles.sh r02
set -u
R=${1:-r02}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 --stream-inputs > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/prof_headline -o h -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --graph 0 --streams 1 > $O/prof_headline.log 2>&1
python3 tools/rocpd_summary.py $O/prof_headline/*results.db > $O/kernel_stats.md
: > $O/other_configs_bench.json
for m in "--model e2evn" "--model e2evn --unet 18x4" "--model qcirim" "--model rvn" "--model ccnn" "--model vsnet" "--rnn GRU --cascades 1 --no-cpu-baseline" "--rnn MGU --cascades 1 --no-cpu-baseline" "--mask 2d --no-cpu-baseline"; do
  python3 bench.py $m --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/other_configs_bench.json
done
python3 bench.py --train --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/train_bench.json
python3 bench.py --train --dtype bf16 --steps 6 --warmup 2 2>/dev/null | tail -1 >> $O/train_bench.json
python3 bench.py --train --model e2evn --steps 6 --warmup 2 2>/dev/null | tail -1 >> $O/train_bench.json
rocprofv3 --kernel-trace --stats -d $O/prof_train -o t -- python3 bench.py --train --dtype bf16 --steps 3 --warmup 1 > $O/prof_train.log 2>&1
python3 tools/rocpd_summary.py $O/prof_train/*results.db > $O/train_bf16_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $O/prof_e2evn -o e -- python3 bench.py --model e2evn --steps 4 --warmup 1 --graph 0 --streams 1 > $O/prof_e2evn.log 2>&1
python3 tools/rocpd_summary.py $O/prof_e2evn/*results.db > $O/e2evn_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $O/prof_2d -o d -- python3 bench.py --mask 2d --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --graph 0 --streams 1 > $O/prof_2d.log 2>&1
python3 tools/rocpd_summary.py $O/prof_2d/*results.db > $O/mask2d_kernel_stats.md
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- python3 tools/probe/pmc_r02.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- python3 tools/probe/pmc_r02.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_m -o p --output-format csv -- python3 tools/probe/pmc_r02.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES -d $O/pmc_c -o p --output-format csv -- python3 tools/probe/pmc_r02.py > /dev/null 2>&1
python3 tools/traffic_json.py $O/pmc_f/*counter_collection.csv $O/pmc_w/*counter_collection.csv $(python3 -c "from mridc_amd import _lib; print(_lib.lib().mrx_version())") $O/pmc_m/*counter_collection.csv $O/pmc_c/*counter_collection.csv > $O/traffic.json
python3 tools/pmc_summary.py $O/pmc_f/*counter_collection.csv $O/pmc_w/*counter_collection.csv $O/pmc_m/*counter_collection.csv $O/pmc_c/*counter_collection.csv > $O/pmc.md
rocprofv3 --kernel-trace --stats -d $O/prof_q -o q -- python3 bench.py --model qcirim --steps 4 --warmup 1 --no-cpu-baseline --graph 0 --streams 1 > $O/prof_q.log 2>&1
python3 tools/rocpd_summary.py $O/prof_q/*results.db > $O/qcirim_kernel_stats.md
rm -rf $O/prof_headline $O/prof_train $O/prof_e2evn $O/prof_2d $O/prof_q
ls -la $O
