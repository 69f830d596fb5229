#!/bin/bash
O=gpurun_out/r04m; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt | cut -c1-250
python __graft_entry__.py --smoke 2>&1 | tail -2
bash tools/runs/r04_profiles.sh v1
