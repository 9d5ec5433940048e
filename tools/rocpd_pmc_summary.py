#!/usr/bin/env python3
"""Per-dispatch-shape PMC table from a rocprofv3 --pmc rocpd database: one row per (kernel, grid) averaged over dispatches.
    python tools/rocpd_pmc_summary.py db [filter-substring]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("mi::", "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace("unsigned short", "bf16")[:70]


def main(path, flt=""):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, counter_name, value, duration, dispatch_id from counters_collection").fetchall()
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    dur = defaultdict(float)
    for kn, gx, gy, gz, wx, cn, val, d, did in rows:
        if flt and flt not in kn:
            continue
        key = (short(kn), gx // max(wx, 1), gy, gz)
        acc[key][cn] += val
        if did not in cnt[key]:
            cnt[key].add(did); dur[key] += d
    names = sorted({cn for v in acc.values() for cn in v})
    print("| kernel | grid | n | us | " + " | ".join(n.replace("SQ_", "") for n in names) + " |")
    print("|---|---|---:|---:|" + "---:|" * len(names))
    for key in sorted(acc, key=lambda k: -dur[k]):
        n = len(cnt[key])
        print("| `%s` | %dx%dx%d | %d | %.1f | " % (key[0], key[1], key[2], key[3], n, dur[key] / n / 1e3) + " | ".join("%.3g" % (acc[key][cn] / n) for cn in names) + " |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
