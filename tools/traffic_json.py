#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the two rocprofv3 --pmc passes over tools/probe/pmc_r02.py (FETCH_SIZE, WRITE_SIZE: per-dispatch means
in KiB).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64
bytes, so it is doubled; WRITE_SIZE is used as is.

    python tools/traffic_json.py FETCH.csv WRITE.csv LIB_VERSION [MFMA_BUSY.csv CU_BUSY.csv] > profiles/r02_traffic.json
"""
import collections
import csv
import json
import sys

KEYS = {            # key in the JSON -> substrings of the kernel names it sums (one "launch" of the operator)
    "llg": ["k_llg372<"],
    "conv_layer1": ["k_rim_layer1_sb", "k_rim_layer<5, 1, 4"],
    "conv_layer2_wino": ["k_rim_layer_wino<0, true, 2, true"],
    "conv_layer2_sb": ["k_rim_layer2_sb<2, true, false, false"],
    "conv_layer2_f16": ["k_rim_layer2_sb<2, true, false, true"],
    "final": ["k_rim_final4"],
    "final_gather": ["k_l2sb_gather"],
    "llg_2d": ["k_fft_rows<false, 1", "k_cols_dc<", "k_cols_dc_t4<", "k_rows_reduce<1", "k_pfa372_expand", "k_pfa372_reduce", "k_llg372_combine"],
}


def per_kernel(path):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]] += float(row["Counter_Value"])
            cnt[row["Kernel_Name"]].add(row["Dispatch_Id"])
    return {k: acc[k] / len(cnt[k]) for k in acc}


def main(fetch_csv, write_csv, lib_version, mfma_csv=None, cubusy_csv=None):
    """mfma_csv / cubusy_csv (optional): passes with SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipe of a SIMD is busy, summed over the
    SIMDs: 32 per v_mfma_f32_32x32x16_bf16) and SQ_BUSY_CU_CYCLES (cycles a CU has work, summed over the CUs) -> per operator
    mfma_util = MFMA busy / (4 SIMDs x CU busy): matrix-pipe utilisation by the hardware counters, at whatever clock the chip sustained."""
    fe, wr = per_kernel(fetch_csv), per_kernel(write_csv)
    mf = per_kernel(mfma_csv) if mfma_csv else {}
    cb = per_kernel(cubusy_csv) if cubusy_csv else {}
    out = {"_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (one counter per pass) -- python3 tools/probe/pmc_r02.py; "
                      "per-dispatch means in KiB",
           "_correction": "gfx950: FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> doubled; WRITE_SIZE as is",
           "lib_version": int(lib_version), "shape": dict(batch=1, coils=15, height=640, width=372, features=64), "kernels": {}}
    for key, pats in KEYS.items():
        names = [k for k in set(fe) | set(wr) if any(p in k for p in pats)]
        if not names:
            continue
        f = sum(fe.get(k, 0.0) for k in names)
        w = sum(wr.get(k, 0.0) for k in names)
        out["kernels"][key] = dict(kernel=" + ".join(sorted(n.split("(")[0][:70] for n in names)), fetch_kib=f, write_kib=w,
                                   hbm_bytes_per_launch=(2.0 * f + w) * 1024.0)
        if mf and cb:
            m, c = sum(mf.get(k, 0.0) for k in names), sum(cb.get(k, 0.0) for k in names)
            out["kernels"][key].update(mfma_busy_cycles=m, cu_busy_cycles=c, mfma_util=(m / (4.0 * c)) if c else None)
    if mf and cb:
        out["_mfma_util"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES and --pmc SQ_BUSY_CU_CYCLES (own passes): mfma_util = MFMA-pipe busy cycles "
                             "/ (4 SIMDs x CU busy cycles), per launch")
        l2 = "conv_layer2_f16" if "conv_layer2_f16" in out["kernels"] else "conv_layer2_sb"     # the default route's kernel
        out["_regulariser"] = ["conv_layer1", l2, "final_gather"]
        reg = [out["kernels"][k] for k in ("conv_layer1", l2, "final_gather") if k in out["kernels"]]
        if len(reg) == 3:
            out["regulariser_mfma_util"] = sum(r["mfma_busy_cycles"] for r in reg) / (4.0 * sum(r["cu_busy_cycles"] for r in reg))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
