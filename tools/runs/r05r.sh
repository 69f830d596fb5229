#!/bin/bash
# round 5, GPU call R: final check of the tree -- GPU suite (product build), GPU suite (CHECK build), smoke, default bench line
O=gpurun_out/r05r; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt | cut -c1-200
MRIDC_AMD_LIB=$PWD/mridc_amd/lib_chk/libmridc_amd.so timeout 900 python -m pytest tests -m gpu -q > $O/pytest_chk.txt 2>&1; tail -2 $O/pytest_chk.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r05r/bench.json') if l.startswith('{')][-1]); print(json.dumps(r['summary'])); print(list(r)[-1], len(json.dumps(r)))
PY
