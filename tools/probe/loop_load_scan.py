"""Loops that wait for their own loads: per kernel of the library (fresh `hipcc -S` output under /tmp/asm5/*.s), the run-time loops whose body issues global /
buffer loads AND reaches `s_waitcnt vmcnt(0)` behind them in the same trip -- every trip pays a memory round trip before its arithmetic, nothing of the
next trip is in flight.  (tools/probe/load_wait_scan.py counts straight-line load -> wait groups; a rolled loop shows there as ONE group however many
trips it makes: k_uconvT's four-planes-per-trip loop, seven trips for a 28-channel layer, was invisible to it.)"""
import glob
import re
import subprocess

rows = []
for path in sorted(glob.glob("/tmp/asm5/*.s")):
    kern, blocks, order = None, {}, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, blocks, order, cur = m.group(1), {}, [], None
            continue
        if kern is None:
            continue
        if line.startswith(".Lfunc_end"):
            idx = {b: i for i, b in enumerate(order)}
            for i, b in enumerate(order):
                for ins in blocks[b]:
                    mm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", ins)
                    if not mm:
                        continue
                    tgt = mm.group(1) or mm.group(2)
                    if tgt in idx and idx[tgt] <= i and i - idx[tgt] <= 6:          # a backward branch over at most a few blocks: a loop
                        body = [x for bb in order[idx[tgt]:i + 1] for x in blocks[bb]]
                        loads = [k for k, x in enumerate(body) if re.match(r"(global_load|buffer_load)", x)]
                        waits = [k for k, x in enumerate(body) if x.startswith("s_waitcnt") and "vmcnt(0)" in x]
                        if loads and any(w > loads[0] for w in waits):
                            work = sum(1 for x in body if x.startswith("v_")), sum(1 for x in body if x.startswith("v_mfma"))
                            rows.append((len(loads), work[0], work[1], path.split("/")[-1][:-2], kern, tgt))
            kern = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        t = line.strip()
        if cur is not None and t and not t.startswith(";") and not t.startswith("."):
            blocks[cur].append(t)
seen = set()
for r in sorted(rows, key=lambda r: -r[1]):
    if (r[4], r[5]) in seen:
        continue
    seen.add((r[4], r[5]))
    dem = subprocess.run(["c++filt", r[4]], capture_output=True, text=True).stdout.strip()
    print(f"{r[0]:3d} loads, {r[1]:4d} vector instr ({r[2]} MFMA) per trip  {r[3]:14s} {dem[:100]}  {r[5]}")
