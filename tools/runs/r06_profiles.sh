#!/bin/bash
# Round-6 profile set (the script of round 5 with round-6 output names): rocprofv3 kernel traces of the headline / E2EVN / qCIRIM / training / 2-D mask runs, PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES: one counter per pass) over tools/probe/pmc_r04.py (+ pmc_r04_b8.py at the bench's batch sizes), then the default bench line.  $1 = output tag.
T=${1:-v1}; O=gpurun_out/r06_$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
prof() {   # tag, bench flags
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$1 -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs $2 > $R/$O/prof_$1.log 2>&1
  python3 $R/tools/rocpd_summary.py $R/$O/prof_$1/t_results.db > $R/$O/${1}_kernel_stats.md 2>/dev/null
  rm -rf $R/$O/prof_$1
  head -12 $R/$O/${1}_kernel_stats.md | cut -c1-140
}
prof headline "--steps 10 --warmup 2"
# the default line keeps two batches of 8 slices in flight: their persistent kernels queue behind one another, so a launch's duration in the trace above
# includes its wait for CUs; the one-stream trace is the one whose averages the bench's HIP-event figures (an instrumented one-stream pass) must agree with
prof headline_one_stream "--steps 10 --warmup 2 --streams 1"
prof e2evn "--model e2evn --steps 4 --warmup 1 --graph 0 --streams 1"
prof qcirim "--model qcirim --steps 6 --warmup 1 --graph 0 --streams 1"
prof train_bf16 "--train --dtype bf16 --steps 3 --warmup 1"
prof mask2d "--mask 2d --steps 6 --warmup 2"
prof e2evn_precision16 "--model e2evn --precision 16 --steps 4 --warmup 1 --graph 0 --streams 1"
prof qcirim_precision16 "--model qcirim --precision 16 --steps 6 --warmup 1 --graph 0 --streams 1"
prof precision16_one_stream "--precision 16 --steps 10 --warmup 2 --streams 1"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$c -o p --output-format csv -- python3 $R/tools/probe/pmc_r04.py > $R/$O/pmc_$c.log 2>&1
done
cd $R
V=$(python -c "from mridc_amd import _lib; print(_lib.lib().mrx_version())")
python tools/traffic_json.py $O/pmc_FETCH_SIZE/*counter_collection.csv $O/pmc_WRITE_SIZE/*counter_collection.csv $V $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc_SQ_BUSY_CU_CYCLES/*counter_collection.csv > $O/traffic.json 2> $O/traffic_json.err
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc8_$c -o p --output-format csv -- python3 $R/tools/probe/pmc_r04_b8.py > $R/$O/pmc8_$c.log 2>&1 )
done
python tools/traffic_json.py $O/pmc8_FETCH_SIZE/*counter_collection.csv $O/pmc8_WRITE_SIZE/*counter_collection.csv $V $O/pmc8_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc8_SQ_BUSY_CU_CYCLES/*counter_collection.csv 8 tools/probe/pmc_r04_b8.py > $O/traffic_b8.json 2>> $O/traffic_json.err
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc4_$c -o p --output-format csv -- python3 $R/tools/probe/pmc_r04_b8.py 4 2d > $R/$O/pmc4_$c.log 2>&1 )
done
python tools/traffic_json.py $O/pmc4_FETCH_SIZE/*counter_collection.csv $O/pmc4_WRITE_SIZE/*counter_collection.csv $V $O/pmc4_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc4_SQ_BUSY_CU_CYCLES/*counter_collection.csv 4 "tools/probe/pmc_r04_b8.py 4 2d" > $O/traffic_b4.json 2>> $O/traffic_json.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
# the default bench line LAST, with this library's counter files in place on the box: `traffic` / `mfma_util_pmc` are reported only for the library version
# the counter passes were made with (bench.measured_traffic)
for s in "" _b8 _b4; do [ -s $O/traffic$s.json ] && cp $O/traffic$s.json profiles/r06_traffic$s.json; done
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
d = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
print("headline", round(d["value"], 2), "streamed", round((d.get("streamed_inputs") or {}).get("value") or 0, 2), "parity", d["parity_vs_oracle"]["rel_l2"], "roofline", round(d["roofline"]["frac"], 3), "fft", round(d["roofline_fft"]["frac"], 3))
for k, v in d["other_configs"].items():
    print("  ", k, v.get("value"), (v.get("parity_vs_oracle") or {}).get("rel_l2"), v.get("error"))
PY
du -sh $O
