#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace rocpd database: every dispatch between the last two Adam kernels with its
queue, start offset, duration, and the idle gap since the previous dispatch on the same queue ended.
    python tools/rocpd_timeline.py db [marker-kernel-substring]"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("mi::", "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace("unsigned short", "bf16")[:64]


def main(path, marker="adam_tf_"):            # adam_tf_kernel (rounds 1-3) / adam_tf_layouts_kernel (round 4)
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = "select %s, start, end, %s, grid_x, grid_y, grid_z from kernels order by start" % (name_col, qcol or "0")
    try:
        rows = c.execute(sel).fetchall()
    except sqlite3.OperationalError:
        rows = [r + (0, 0, 0) for r in c.execute("select %s, start, end, %s from kernels order by start" % (name_col, qcol or "0")).fetchall()]
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        print("marker kernel not found twice"); return
    lo, hi = marks[-2] + 1, marks[-1] + 1
    t0 = rows[lo][1]
    last_end = {}
    busy = 0.0
    print("step wall %.1f us (end of previous %s -> end of this one)" % ((rows[hi - 1][2] - rows[lo - 1][2]) / 1e3, marker))
    print("| # | queue | start us | dur us | gap us | kernel | grid |")
    print("|---:|---:|---:|---:|---:|---|---|")
    for i in range(lo, hi):
        n, s, e, q, gx, gy, gz = rows[i]
        gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = e
        print("| %d | %s | %.1f | %.1f | %.1f | `%s` | %sx%sx%s |" % (i - lo, q, (s - t0) / 1e3, (e - s) / 1e3, gap, short(n), gx, gy, gz))


if __name__ == "__main__":
    main(*sys.argv[1:3])
