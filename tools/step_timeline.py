#!/usr/bin/env python
"""Kernel sequence of ONE steady-state training step out of a rocprofv3 --kernel-trace rocpd database: the dispatches between the
last two optimizer launches (wsi::adam_step_kernel or torch's fused Adam), in start order, with duration and the idle gap in front of each.  `python tools/step_timeline.py x.db [out.txt]`"""
import re
import sqlite3
import sys


def main(path, out=None):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if "FusedAdam" in r[0] or "adam_step_kernel" in r[0]]
    # two multi_tensor launches per step: the step is what lies between the last launch of step k-1 and the last of step k
    ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] != i + 1]
    a, b = ends[-2], ends[-1]
    lines, prev, busy = [], rows[a][2], 0.0
    for n, s, e in rows[a + 1:b + 1]:
        n = re.sub(r"^void ", "", n)
        lines.append(f"{(s - prev) / 1e3:7.1f} {(e - s) / 1e3:8.1f}  {n[:130]}")
        busy += (e - s) / 1e3
        prev = max(prev, e)
    wall = (rows[b][2] - rows[a][2]) / 1e3
    head = f"# one step: {b - a} dispatches, wall {wall:.1f} us, kernel time {busy:.1f} us\n#  gap_us   dur_us  kernel"
    text = head + "\n" + "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
