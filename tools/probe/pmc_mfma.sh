#!/bin/bash
# MFMA-pipe busy cycles and the clock the chip sustains, per kernel of the headline loop (one counter per rocprofv3 pass).
set -u
O=gpurun_out/${1:-pmc_mfma}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16; do
  rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python3 tools/probe/pmc_r02.py > $O/$c.log 2>&1
  python3 - $O/$c/*counter_collection.csv $c <<'PY'
import csv, sys, collections
acc, cnt = collections.defaultdict(float), collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:60]
    acc[k] += float(row["Counter_Value"]); cnt[k].add(row["Dispatch_Id"])
for k in sorted(acc):
    if any(s in k for s in ("k_rim_layer", "k_llg372<", "k_l2sb_gather", "k_cols_dc_t4")):
        print(sys.argv[2], k, acc[k] / len(cnt[k]))
PY
done
python3 - $O/SQ_VALU_MFMA_BUSY_CYCLES/*kernel_trace.csv <<'PY'
import csv, sys, collections
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:60]
    acc[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); cnt[k] += 1
for k in sorted(acc):
    if any(s in k for s in ("k_rim_layer", "k_llg372<", "k_l2sb_gather", "k_cols_dc_t4")):
        print("wall_ns(profiled)", k, acc[k] / cnt[k])
PY
