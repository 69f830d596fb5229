#!/bin/bash
O=gpurun_out/r04ad13; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
bash tools/runs/r04_profiles.sh v13
