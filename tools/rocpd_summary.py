#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output of `rocprofv3 --kernel-trace --stats`) as a
per-kernel table: calls, total / average / min / max duration, share of GPU kernel time.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("mi::", "")
    name = re.sub(r"\(.*$", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        d = (e - s) / 1e3
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.2f |" % (short(n), a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / total))
    print("\ntotal kernel time: %.1f us over %d dispatches" % (total, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1])
