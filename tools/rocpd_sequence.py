#!/usr/bin/env python3
"""Per-dispatch view of a rocprofv3 rocpd database: the repeating launch sequence that ends the trace (one rollout step, one SGD step ...),
averaged over its last `reps` repetitions: per position the kernel, its duration, and the gap since the previous kernel ended.
    python tools/rocpd_sequence.py x_results.db <kernels per repetition> [reps]"""
import sqlite3
import sys
import numpy as np
from rocpd_summary import short


def main(path, n, reps=100):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    rows = rows[-n * reps:]
    names = [short(r[0]) for r in rows[-n:]]
    s = np.array([r[1] for r in rows], float).reshape(reps, n); e = np.array([r[2] for r in rows], float).reshape(reps, n)
    dur = (e - s) / 1e3
    gap = np.zeros_like(dur); gap[:, 1:] = (s[:, 1:] - e[:, :-1]) / 1e3
    print("| # | kernel | median us | min us | gap before, median us |\n|---:|---|---:|---:|---:|")
    for i in range(n):
        print("| %d | `%s` | %.2f | %.2f | %.2f |" % (i, names[i], np.median(dur[:, i]), dur[:, i].min(), np.median(gap[:, i])))
    span = (e[:, -1] - s[:, 0]) / 1e3
    print("\nfirst start -> last end: median %.1f us, min %.1f us; sum of durations %.1f us" % (np.median(span), span.min(), np.median(dur.sum(1))))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 100)
