#!/bin/bash
# batch-8 counter passes -> profiles/r04_traffic_b8.json (on the box, so that the bench of the same call reports it), default bench line, headline kernel trace
O=gpurun_out/r04ax; mkdir -p $O; R=$GRAFT_REPO_ROOT
V=$(python -c "from mridc_amd import _lib; print(_lib.lib().mrx_version())")
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc8_$c -o p --output-format csv -- python3 $R/tools/probe/pmc_r04_b8.py > $R/$O/pmc8_$c.log 2>&1 )
done
python tools/traffic_json.py $O/pmc8_FETCH_SIZE/*counter_collection.csv $O/pmc8_WRITE_SIZE/*counter_collection.csv $V $O/pmc8_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc8_SQ_BUSY_CU_CYCLES/*counter_collection.csv 8 tools/probe/pmc_r04_b8.py > $O/traffic_b8.json 2> $O/traffic_json.err
python tools/pmc_summary.py $O/pmc8_*/*counter_collection.csv > $O/pmc_b8.md 2>/dev/null
cp $O/traffic_b8.json profiles/r04_traffic_b8.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04ax/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], "streamed", (d.get("streamed_inputs") or {}).get("value"), "traffic", d["roofline"]["traffic"], d["roofline_fft"]["traffic"], d["roofline_fft"]["gather_form"]["traffic"], "mfma", d["roofline"].get("mfma_util_pmc"), "frac", d["roofline"]["frac"], d["roofline_fft"]["frac"])
for k, v in d["other_configs"].items(): print("  ", k, v.get("value"), (v.get("roofline") or {}).get("traffic"))
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --model e2evn --steps 4 --warmup 1 --graph 0 --streams 1 > $R/$O/prof.log 2>&1
python3 $R/tools/rocpd_summary.py $R/$O/prof/t_results.db > $R/$O/e2evn_kernel_stats.md 2>/dev/null
rm -rf $R/$O/prof
head -9 $R/$O/e2evn_kernel_stats.md | cut -c1-150
