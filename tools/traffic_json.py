#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the two rocprofv3 --pmc passes over tools/probe/pmc_r02.py (FETCH_SIZE, WRITE_SIZE: per-dispatch means
in KiB).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64
bytes, so it is doubled; WRITE_SIZE is used as is.

    python tools/traffic_json.py FETCH.csv WRITE.csv LIB_VERSION [MFMA_BUSY.csv CU_BUSY.csv [BATCH PROBE]] > profiles/r02_traffic.json
(BATCH: slices per launch of the probe -- pmc_r04_b8.py runs the headline loop's kernels at the bench's default 8 -> profiles/r04_traffic_b8.json)
"""
import collections
import csv
import json
import sys

KEYS = {            # key in the JSON -> substrings of the kernel names it sums (one "launch" of the operator)
    "llg": ["k_llg372<0, true, false>"],      # (pmc_r04.py runs the default form: the data term as a constant plane; <0, false, false> = that plane's one launch per slice)
    "llg_gather": ["k_llg372<0, true, true>"],                             # ... with the previous step's nine-tap gather folded in
    "conv_layer1": ["k_rim_layer1_sb", "k_rim_layer<5, 1, 4"],
    "conv_layer2_wino": ["k_rim_layer_wino<0, true, 2, true"],
    "conv_layer2_sb": ["k_rim_layer2_sb<2, true, false, false"],
    "conv_layer2_f16": ["k_rim_layer2_sb<2, true, false, true"],
    "final": ["k_rim_final4"],
    "amp_layer1": ["k_amp_layer1_t"],          # round 6: the precision-16 route (csrc/rim_amp16.hip)
    "amp_layer2": ["k_amp_layer2"],
    "final_gather": ["k_l2sb_gather"],
    "llg_2d": ["k_fft_rows<false, 1", "k_cols_dc<", "k_cols_dc_t4<", "k_rows_reduce<1", "k_pfa372_expand", "k_pfa372_reduce", "k_llg372_combine"],   # (r04: the deferred form without y)
    # round 4 (tools/probe/pmc_r04.py): dominant kernels of the other configurations, each at its own shape (`at`)
    "llg_2d_cols_noy": ["k_cols_dc_t4<PlanCT<640, 5, 8, 4, 4>, true>"],
    # (round 5 added a fourth template argument: the ticket form; round 6 a fifth and a sixth: rows per work item -- 16 at the bench's batch of 8, 8 at the batch-4 probe -- and fp16 terms)
    "e2evn_uconv_h_14to14": ["k_uconv_h<1, 1, true>", "k_uconv_h<1, 1, true, false>", "k_uconv_h<1, 1, true, false, 16, 2>", "k_uconv_h<1, 1, true, false, 8, 2>"],
    "qcirim_conv3x3_h_128": ["k_uconv_h<4, 2, false>", "k_uconv_h<4, 2, false, false>", "k_uconv_h<4, 2, false, false, 8, 2>"],
    "train_layer2_fwd": ["k_conv_bf16<3, 2, 64, 2, 0, 2>", "k_conv_bf16<3, 2, 64, 2, 2, 2>"],
    "train_cell_bwd": ["k_tl_cell_bwd<true, true>", "k_tl_cell_bwd<true, true, true>"],      # (r04 lib 243+: the state as its mask words)
    "train_wgrad_3x3d2": ["k_conv_wgrad_bf16<3, 2, 1>", "k_conv_wgrad_bf16<3, 2, 1, 1>"],
    "train_dgrad_3x3d2": ["k_conv_bf16<3, 2, 64, 2, 1, 1>", "k_tl_dgrad64<3, 2, false>"],      # (lib 248+: weights resident in LDS)
}
AT = {"e2evn_uconv_h_14to14": "4 x 14 -> 14 x 640 x 380", "qcirim_conv3x3_h_128": "1 x 128 -> 128 x 256 x 256, dilation 2",
      "train_layer2_fwd": "1 x 64 x 640 x 372", "train_cell_bwd": "1 x 64 x 640 x 372", "train_wgrad_3x3d2": "1 x 64 x 640 x 372",
      "train_dgrad_3x3d2": "1 x 64 x 640 x 372", "llg_2d_cols_noy": "15 x 640 x 372", "llg_gather": "15 x 640 x 372"}


def per_kernel(path):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]] += float(row["Counter_Value"])
            cnt[row["Kernel_Name"]].add(row["Dispatch_Id"])
    return {k: acc[k] / len(cnt[k]) for k in acc}


def main(fetch_csv, write_csv, lib_version, mfma_csv=None, cubusy_csv=None, batch=1, probe="tools/probe/pmc_r04.py"):
    """mfma_csv / cubusy_csv (optional): passes with SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipe of a SIMD is busy, summed over the
    SIMDs: 32 per v_mfma_f32_32x32x16_bf16) and SQ_BUSY_CU_CYCLES (cycles a CU has work, summed over the CUs) -> per operator
    mfma_util = MFMA busy / (4 SIMDs x CU busy): matrix-pipe utilisation by the hardware counters, at whatever clock the chip sustained."""
    fe, wr = per_kernel(fetch_csv), per_kernel(write_csv)
    mf = per_kernel(mfma_csv) if mfma_csv else {}
    cb = per_kernel(cubusy_csv) if cubusy_csv else {}
    out = {"_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (one counter per pass) -- python3 " + probe + "; "
                      "per-dispatch means in KiB",
           "_correction": "gfx950: FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> doubled; WRITE_SIZE as is.  The guide calibrated "
                          "that factor on 16-byte-per-lane streams; the round-4 training kernels load 4 bytes per lane from 64 planes -- for k_tl_cell_bwd, whose "
                          "244 MB of reads have no reuse, the UNDOUBLED counter (234 MB) is the one that matches, so `hbm_bytes_uncorrected` (FETCH + WRITE) is kept "
                          "beside the prescribed figure for every kernel",
           "lib_version": int(lib_version), "shape": dict(batch=int(batch), coils=15, height=640, width=372, features=64), "kernels": {}}
    for key, pats in KEYS.items():
        names = [k for k in set(fe) | set(wr) if any(p in k for p in pats)]
        if not names:
            continue
        f = sum(fe.get(k, 0.0) for k in names)
        w = sum(wr.get(k, 0.0) for k in names)
        out["kernels"][key] = dict(kernel=" + ".join(sorted(n.split("(")[0][:70] for n in names)), fetch_kib=f, write_kib=w,
                                   hbm_bytes_per_launch=(2.0 * f + w) * 1024.0, hbm_bytes_uncorrected=(f + w) * 1024.0)
        if key in AT and int(batch) == 1:
            out["kernels"][key]["at"] = AT[key]
        elif key == "e2evn_uconv_h_14to14":
            out["kernels"][key]["at"] = f"{int(batch)} x 14 -> 14 x 640 x 380"
        if mf and cb:
            m, c = sum(mf.get(k, 0.0) for k in names), sum(cb.get(k, 0.0) for k in names)
            out["kernels"][key].update(mfma_busy_cycles=m, cu_busy_cycles=c, mfma_util=(m / (4.0 * c)) if c else None)
    if mf and cb:
        out["_mfma_util"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES and --pmc SQ_BUSY_CU_CYCLES (own passes): mfma_util = MFMA-pipe busy cycles "
                             "/ (4 SIMDs x CU busy cycles), per launch")
        l2 = "conv_layer2_f16" if "conv_layer2_f16" in out["kernels"] else "conv_layer2_sb"     # the default route's kernel
        # launches of the regulariser per RIM step: layer 1, layer 2, and (lib 242+) one stand-alone gather per CASCADE of eight steps -- the other seven
        # ride in the next step's gradient launch; their extra CU-busy cycles there (gather form minus plain form of the gradient kernel) are charged here
        out["_regulariser"] = ["conv_layer1", l2, "final_gather / 8", "7/8 (llg_gather - llg)"]
        ks = out["kernels"]
        if all(k in ks for k in ("conv_layer1", l2, "final_gather")):
            mfma = ks["conv_layer1"]["mfma_busy_cycles"] + ks[l2]["mfma_busy_cycles"] + ks["final_gather"]["mfma_busy_cycles"] / 8.0
            cu = ks["conv_layer1"]["cu_busy_cycles"] + ks[l2]["cu_busy_cycles"] + ks["final_gather"]["cu_busy_cycles"] / 8.0
            if "llg_gather" in ks and "llg" in ks:
                cu += 7.0 / 8.0 * max(ks["llg_gather"]["cu_busy_cycles"] - ks["llg"]["cu_busy_cycles"], 0.0)
            else:
                cu += 7.0 / 8.0 * ks["final_gather"]["cu_busy_cycles"]
            out["regulariser_mfma_util"] = mfma / (4.0 * cu)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:8])
