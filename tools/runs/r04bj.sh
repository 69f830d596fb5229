#!/bin/bash
O=gpurun_out/r04bj; mkdir -p $O
timeout 330 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
