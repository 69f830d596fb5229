"""Join a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE run of tools/probe/l2_power.py (PROBE_REPS=1 PROBE_N=20: five cases x 26 launches of the layer-2 kernel,
in order) on the dispatch id: per case the mean launch duration and the effective shader clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration."""
import csv
import glob
import sys

d = sys.argv[1]
per_case = int(sys.argv[2]) if len(sys.argv) > 2 else 26
dur, cnt = {}, {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_rim_layer2_sb" in row["Kernel_Name"]:
            dur[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_rim_layer2_sb" in row["Kernel_Name"] and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[int(row["Dispatch_Id"])] = cnt.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
ids = sorted(i for i in dur if i in cnt)
names = ["random states, random weights", "states exact in fp16 (low terms zero)", "states and weights exact in fp16", "zero states, random weights", "zero states, zero weights"]
print(f"{len(ids)} launches of k_rim_layer2_sb with a duration and a counter value")
for c in range(len(ids) // per_case):
    sel = ids[c * per_case + 6:(c + 1) * per_case]                   # (the six warm-up launches dropped)
    t = sum(dur[i] for i in sel) / len(sel)
    g = sum(cnt[i] for i in sel) / len(sel)
    print(f"{names[c % 5]:42s} {t / 8:7.2f} us per slice   GRBM_GUI_ACTIVE {g:12.0f}   effective clock {g / 8 / t / 1e3:5.2f} GHz (/8 XCDs)   {g / t / 1e3:5.2f} (raw)")
