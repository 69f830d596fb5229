#!/bin/bash
# round 5, GPU call D: where do the dominant layer's waves spend their cycles?  PMC passes over base and newst_rows2 (l2_pmc.py: 6 launches of each layer)
O=gpurun_out/r05d; mkdir -p $O; R=$PWD
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"
P4="SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT"
for v in base newst_rows2; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    ( cd /tmp && MRIDC_AMD_LIB=$R/mridc_amd/lib_v_$v/libmridc_amd.so timeout 300 rocprofv3 --kernel-trace --pmc $P -d $R/$O/${v}_p$i -o p --output-format csv -- python3 $R/tools/probe/l2_pmc.py > $R/$O/${v}_p$i.log 2>&1 )
  done
  echo "== $v" >> $O/pmc_summary.txt
  python tools/probe/pmc_sum.py $O/${v}_p1 $O/${v}_p2 $O/${v}_p3 $O/${v}_p4 >> $O/pmc_summary.txt 2>&1
  for i in 1 2 3 4; do python - <<PY >> $O/pmc_summary.txt
import csv,glob
for f in glob.glob("$O/${v}_p$i/**/*kernel_trace.csv", recursive=True):
    d=[(r["Kernel_Name"][:40], int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "rim_layer2" in r["Kernel_Name"]]
    print("pass $i layer2 durations ns:", [x[1] for x in d][-3:])
PY
  done
  rm -rf $O/${v}_p*/*/*.db 2>/dev/null
done
cat $O/pmc_summary.txt | head -150
