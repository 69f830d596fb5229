"""Sum rocprofv3 counter_collection CSVs per (kernel, counter): python tools/probe/pmc_sum.py DIR...  -> mean counter value per launch of each kernel."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:60]
            per[(row["Dispatch_Id"], k, row["Counter_Name"])] += float(row["Counter_Value"])
        for (_, k, c), v in per.items():
            acc[k][c].append(v)
for k in sorted(acc):
    if "rim_layer" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        v = v[len(v) // 2:]                 # the later (warm) launches
        print(f"   {c:32s} {sum(v) / len(v):16.0f}   ({len(v)} launches)")
