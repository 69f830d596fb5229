#!/bin/bash
# counter passes of the 2-D-mask loop at its default batch of 4 -> profiles/r04_traffic_b4.json (on the box), then the default bench line
O=gpurun_out/r04ay; mkdir -p $O; R=$GRAFT_REPO_ROOT
V=$(python -c "from mridc_amd import _lib; print(_lib.lib().mrx_version())")
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc4_$c -o p --output-format csv -- python3 $R/tools/probe/pmc_r04_b8.py 4 2d > $R/$O/pmc4_$c.log 2>&1 )
done
python tools/traffic_json.py $O/pmc4_FETCH_SIZE/*counter_collection.csv $O/pmc4_WRITE_SIZE/*counter_collection.csv $V $O/pmc4_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc4_SQ_BUSY_CU_CYCLES/*counter_collection.csv 4 "tools/probe/pmc_r04_b8.py 4 2d" > $O/traffic_b4.json 2> $O/traffic_json.err
python tools/pmc_summary.py $O/pmc4_*/*counter_collection.csv > $O/pmc_b4.md 2>/dev/null
cp $O/traffic_b4.json profiles/r04_traffic_b4.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04ay/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], "streamed", (d.get("streamed_inputs") or {}).get("value"), "traffic", d["roofline"]["traffic"], "frac", d["roofline"]["frac"], d["roofline_fft"]["frac"])
for k, v in d["other_configs"].items(): print("  ", k, v.get("value"), (v.get("roofline") or {}).get("traffic"), (v.get("roofline_fft") or {}).get("traffic"))
PY
tail -3 $O/traffic_json.err
