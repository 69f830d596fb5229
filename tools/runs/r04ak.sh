#!/bin/bash
# PMC passes + traffic table + the default line (after a probe / bench change that does not touch the library)
O=gpurun_out/r04_v5; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  rm -rf $R/$O/pmc_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$c -o p --output-format csv -- python3 $R/tools/probe/pmc_r04.py > $R/$O/pmc_$c.log 2>&1
done
cd $R
V=$(python -c "from mridc_amd import _lib; print(_lib.lib().mrx_version())")
python tools/traffic_json.py $O/pmc_FETCH_SIZE/*counter_collection.csv $O/pmc_WRITE_SIZE/*counter_collection.csv $V $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc_SQ_BUSY_CU_CYCLES/*counter_collection.csv > $O/traffic.json 2> $O/traffic_json.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cp $O/traffic.json profiles/r04_traffic.json
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_v5/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], "traffic", d["roofline"]["traffic"], d["roofline_fft"]["traffic"])
for k, v in d["other_configs"].items(): print("  ", k, v.get("value"), (v.get("roofline") or {}).get("traffic"), (v.get("roofline") or {}).get("mfma_util_pmc"))
PY
