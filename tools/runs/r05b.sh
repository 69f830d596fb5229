#!/bin/bash
# round 5, GPU call B: DUNet reproducer (30 x, one process), the GPU suite on the CHECK build (no -x), the CPU-oracle thread sweep, a bench line
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python tools/probe/dunet_repro.py 30 > $O/repro.txt 2>&1; grep "^part" $O/repro.txt | tail -3
MRIDC_AMD_LIB=$PWD/mridc_amd/lib_chk/libmridc_amd.so timeout 900 python -m pytest tests -m gpu -q > $O/pytest_chk.txt 2>&1; tail -3 $O/pytest_chk.txt | cut -c1-300
timeout 900 python tools/probe/cpu_thread_sweep.py > $O/cpu_thread_sweep.txt 2>&1; tail -8 $O/cpu_thread_sweep.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r05b/bench.json') if l.startswith('{')][-1]); print(json.dumps(r['summary']))
PY
