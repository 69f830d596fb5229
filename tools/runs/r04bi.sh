#!/bin/bash
# lib 256 (k_pfa372_expand<RED, DC>): the tests that reach the W = 372 expand pass, the three counter probes -> profiles/r04_traffic*.json (on the box), the default
# bench line, a one-stream trace of the 2-D-mask line
O=gpurun_out/r04bi; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 420 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_headline.py tests/test_gpu_backward.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt | cut -c1-200
V=$(python -c "from mridc_amd import _lib; print(_lib.lib().mrx_version())")
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$c -o p --output-format csv -- python3 $R/tools/probe/pmc_r04.py > $R/$O/pmc_$c.log 2>&1 )
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc8_$c -o p --output-format csv -- python3 $R/tools/probe/pmc_r04_b8.py > $R/$O/pmc8_$c.log 2>&1 )
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc4_$c -o p --output-format csv -- python3 $R/tools/probe/pmc_r04_b8.py 4 2d > $R/$O/pmc4_$c.log 2>&1 )
done
python tools/traffic_json.py $O/pmc_FETCH_SIZE/*counter_collection.csv $O/pmc_WRITE_SIZE/*counter_collection.csv $V $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc_SQ_BUSY_CU_CYCLES/*counter_collection.csv > $O/traffic.json 2> $O/traffic_json.err
python tools/traffic_json.py $O/pmc8_FETCH_SIZE/*counter_collection.csv $O/pmc8_WRITE_SIZE/*counter_collection.csv $V $O/pmc8_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc8_SQ_BUSY_CU_CYCLES/*counter_collection.csv 8 tools/probe/pmc_r04_b8.py > $O/traffic_b8.json 2>> $O/traffic_json.err
python tools/traffic_json.py $O/pmc4_FETCH_SIZE/*counter_collection.csv $O/pmc4_WRITE_SIZE/*counter_collection.csv $V $O/pmc4_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc4_SQ_BUSY_CU_CYCLES/*counter_collection.csv 4 "tools/probe/pmc_r04_b8.py 4 2d" > $O/traffic_b4.json 2>> $O/traffic_json.err
for t in "" 8 4; do s=${t:+_b$t}; python tools/pmc_summary.py $O/pmc${t}_FETCH_SIZE/*counter_collection.csv $O/pmc${t}_WRITE_SIZE/*counter_collection.csv $O/pmc${t}_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc${t}_SQ_BUSY_CU_CYCLES/*counter_collection.csv > $O/pmc$s.md 2>/dev/null; cp $O/traffic$s.json profiles/r04_traffic$s.json; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
( time timeout 600 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04bi/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], "traffic", d["roofline"]["traffic"], d["roofline_fft"]["traffic"], "mfma", d["roofline"].get("mfma_util_pmc"))
for k, v in d["other_configs"].items(): print("  ", k, v.get("value"), (v.get("roofline") or {}).get("traffic"), (v.get("roofline") or {}).get("avg_ms"))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --mask 2d --streams 1 --steps 6 --warmup 2 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/mask2d_one_stream_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
sed -n 5,9p $R/$O/mask2d_one_stream_kernel_stats.md | cut -c1-150
