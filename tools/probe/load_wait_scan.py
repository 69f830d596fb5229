"""Counts, per kernel of the library, the groups of global loads that are followed by `s_waitcnt vmcnt(0)` before another load goes out ("short" =
groups of at most two loads): the signature of loads waited for one by one -- a load sunk next to a use that sits behind a branch, a run-time loop of
load -> LDS write, a read-modify-write per element.  Round 4 (libs 252-255) found the gather inside k_llg372, k_pfa372_reduce, k_cols_dc_t4,
k_pfa372_expand's data-consistency epilogue, k_conv1x1_sb128, k_conv_sbs, k_uconv_h<.., false>, k_tl_wgrad_in and the cell backward's slot update with it.

    for f in mridc_amd/csrc/*.hip; do b=$(basename $f .hip); mkdir -p /tmp/asm/$b; cp $f /tmp/asm/$b/x.hip;
      (cd /tmp/asm/$b && hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMRX_NO_PACKED_FP32 -Xclang -target-feature -Xclang -packed-fp32-ops \
          -I$PWD/mridc_amd/csrc -x hip --cuda-device-only -S x.hip -o x.s); done
    python tools/probe/load_wait_scan.py
"""
import re, sys, glob, subprocess
# per kernel: number of "serialised" load groups = vmcnt(0) waits that follow a global load with no other load issued after... simple metric:
# count sequences LD+ -> vmcnt(0) ; report kernels with many such sequences (>=4)
rows = []
for path in sorted(glob.glob("/tmp/asm/*/x.s")):
    name = None; seq = 0; loads = 0; pending = 0; nload_total = 0; single = 0
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1); seq = 0; nload_total = 0; pending = 0; single = 0
            continue
        if name is None: continue
        if line.startswith(".Lfunc_end"):
            if seq >= 4:
                rows.append((single, seq, nload_total, path.split("/")[3], name))
            name = None; continue
        if re.search(r"\b(global_load|buffer_load)", line):
            pending += 1; nload_total += 1
        elif "s_waitcnt" in line and "vmcnt(0)" in line:
            if pending:
                seq += 1
                if pending <= 2: single += 1
            pending = 0
rows.sort(reverse=True)
for r in rows[:45]:
    dem = subprocess.run(["c++filt", r[4]], capture_output=True, text=True).stdout.strip()
    print(f"{r[0]:4d} short / {r[1]:4d} groups / {r[2]:4d} loads  {r[3]:16s} {dem[:110]}")
