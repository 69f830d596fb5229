#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter CSVs (one file per pass).

    python tools/pmc_summary.py gpurun_out/pmc/p1_counter_collection.csv [more.csv ...]
"""
import collections
import csv
import sys


def main(paths):
    for path in paths:
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(set)
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"][:60]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k].add(row["Dispatch_Id"])
        names = sorted({c for k in acc for c in acc[k]})
        print(f"## {path}\n")
        print("| kernel | dispatches | " + " | ".join(names) + " |")
        print("|---|---:|" + "---:|" * len(names))
        for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
            n = len(cnt[k])
            print(f"| `{k}` | {n} | " + " | ".join(f"{acc[k][c] / n:.4g}" for c in names) + " |")
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
