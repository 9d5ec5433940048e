#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing, per basic block (label), to see where VALU / SALU / LDS / VMEM / MFMA
instructions sit (tools for the VALU-per-MFMA work of round 2).   usage: asm_stats.py file.s <mangled-name-substring> [--blocks]"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, sub = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if sub in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\S+:", l) and sub in l.split(":")[0]:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    blocks, cur, name = [], {}, "entry"
    order = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            blocks.append((name, cur)); cur = {}; name = m.group(1)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        c = classify(op)
        cur[c] = cur.get(c, 0) + 1
        if c == "vmem" and "lds" in t:
            cur["ldsdma"] = cur.get("ldsdma", 0) + 1
    blocks.append((name, cur))
    tot = {}
    for n, b in blocks:
        for k, v in b.items():
            tot[k] = tot.get(k, 0) + v
    print("total", dict(sorted(tot.items())))
    if show_blocks:
        for n, b in blocks:
            if sum(b.values()) >= 20:
                print("%-14s" % n, dict(sorted(b.items())))
    for l in lines[start:]:
        if ".vgpr_count" in l or ".sgpr_count" in l or "vgpr_spill" in l or "; Occupancy" in l or "; ScratchSize" in l or "; NumVgprs" in l or "; NumAgprs" in l:
            print(l.strip())
        if l.startswith(".Lfunc_end"):
            pass
        if "; Occupancy" in l:
            break


main()
