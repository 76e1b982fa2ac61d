#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite result (``--kernel-trace``) into a per-kernel stats table
(name, calls, total/avg/min/max µs, % of GPU kernel time) — the `--stats` view, as text/CSV for profiles/."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 150 else name[:147] + "..."


def main(path, out=None, skip_first=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    agg = {}
    for n, s, e in rows:
        d = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        us = (e - s) / 1e3
        d[0] += 1
        d[1] += us
        d[2] = min(d[2], us)
        d[3] = max(d[3], us)
    total = sum(v[1] for v in agg.values())
    lines = ["name,calls,total_us,avg_us,min_us,max_us,pct"]
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{short(n)}\",{v[0]},{v[1]:.1f},{v[1] / v[0]:.2f},{v[2]:.2f},{v[3]:.2f},{100 * v[1] / total:.2f}")
    lines.append(f"\"TOTAL\",{sum(v[0] for v in agg.values())},{total:.1f},,,,100.00")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
