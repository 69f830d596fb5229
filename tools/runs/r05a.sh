#!/bin/bash
# round 5, first GPU call: the DUNet reproducer (30 x in one process), the whole GPU suite on the CHECK build, then on the product build + smoke
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python tools/probe/dunet_repro.py 30 > $O/repro.txt 2>&1; grep "^part" $O/repro.txt | tail -4
MRIDC_AMD_LIB=$PWD/mridc_amd/lib_chk/libmridc_amd.so timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_chk.txt 2>&1; tail -3 $O/pytest_chk.txt | cut -c1-300
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
