"""Effective shader clock per kernel = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / launch duration, from a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE
output directory (csv): python tools/probe/kernel_clocks.py DIR"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
dur, cnt, name = {}, {}, {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        i = int(row["Dispatch_Id"])
        dur[i] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        name[i] = row["Kernel_Name"].split("(")[0][:70]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            i = int(row["Dispatch_Id"])
            cnt[i] = cnt.get(i, 0.0) + float(row["Counter_Value"])
acc = defaultdict(list)
for i in dur:
    if i in cnt and dur[i] > 0:
        acc[name[i]].append((dur[i], cnt[i]))
print(f"{'kernel':72s} {'launches':>8s} {'us':>9s} {'GHz':>6s}")
for k, v in sorted(acc.items(), key=lambda kv: -sum(t for t, _ in kv[1])):
    v = v[len(v) // 2:]
    t = sum(a for a, _ in v) / len(v)
    g = sum(b for _, b in v) / len(v)
    if t >= 8.0:
        print(f"{k:72s} {len(v):8d} {t:9.1f} {g / 8 / t / 1e3:6.2f}")
