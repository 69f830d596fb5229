#!/bin/bash
# round 4, GPU call L: side-stream tape, fixed general-mask test, PMC passes (traffic + matrix-pipe busy) over tools/probe/pmc_r04.py
O=gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_train_bf16.py tests/test_gpu_backward.py -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-stream-inputs --train --dtype bf16 --steps 4 --warmup 1 > $O/bench_train.json 2> $O/bench_train.err; head -c 250 $O/bench_train.json; echo
timeout 300 python tools/probe/tl_determinism.py 15 640 372 > $O/determinism.txt 2>&1; tail -4 $O/determinism.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probe/pmc_r04.py > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/$O/pmc_$c.log
done
cd $GRAFT_REPO_ROOT
V=$(python -c "from mridc_amd import _lib; print(_lib.lib().mrx_version())")
python tools/traffic_json.py $O/pmc_FETCH_SIZE/*counter_collection.csv $O/pmc_WRITE_SIZE/*counter_collection.csv $V $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/*counter_collection.csv $O/pmc_SQ_BUSY_CU_CYCLES/*counter_collection.csv > $O/r04_traffic.json 2> $O/traffic_json.err
python - <<'PY'
import json
t = json.load(open("gpurun_out/r04l/r04_traffic.json"))
for k, v in t["kernels"].items():
    print(f"{k:26s} {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB  mfma_util {v.get('mfma_util')}")
print("regulariser", t.get("regulariser_mfma_util"))
PY
find $O -name "*kernel_trace.csv" -size +20M -delete; du -sh $O
