"""Per-kernel means of the rocprofv3 --pmc CSVs of tools/probe/llg2d_passes.py: python3 llg2d_passes_summary.py TAG=counter_collection.csv ..."""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for arg in sys.argv[1:]:
    tag, path = arg.split("=", 1)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(path) as f:
        for r in csv.DictReader(f):
            per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), c in per.items():
        if not any(s in k for s in ("k_pfa372_expand", "k_cols_dc", "k_pfa372_reduce")):
            continue
        for name, v in c.items():
            rows[(tag, k.split("(")[0][:60])][name].append(v)
names = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"]
print("| batch | kernel | " + " | ".join(n.replace("SQ_", "") for n in names) + " | VALU / wave cycles | parked (WAIT_ANY) / wave cycles | issue-stalled / wave cycles | waves per SIMD (WAVE / 4 BUSY_CU) |")
print("|---|---|" + "---:|" * (len(names) + 4))
for (tag, k), c in sorted(rows.items()):
    m = {n: (sum(c[n]) / len(c[n]) if c.get(n) else None) for n in names}
    wc = m["SQ_WAVE_CYCLES"]
    f = lambda v: "-" if v is None else f"{v:.4g}"  # noqa: E731
    rat = lambda a: "-" if (m[a] is None or not wc) else f"{m[a] / wc:.3f}"  # noqa: E731
    occ = "-" if (not wc or not m["SQ_BUSY_CU_CYCLES"]) else f"{wc / (4 * m['SQ_BUSY_CU_CYCLES']):.2f}"
    print(f"| {tag} | `{k}` | " + " | ".join(f(m[n]) for n in names) + f" | {rat('SQ_ACTIVE_INST_VALU')} | {rat('SQ_WAIT_ANY')} | {rat('SQ_WAIT_INST_ANY')} | {occ} |")
