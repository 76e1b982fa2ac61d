#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately, as the
MI355X guide prescribes).  Units: counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced streaming reads, so the read side is doubled (MI355X_MICROARCH.md §HBM)."""
import collections
import csv
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main(fetch_csv, write_csv, out=None):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    lines = ["kernel,launches,fetch_MB_per_launch(x2 corrected),write_MB_per_launch,hbm_MB_per_launch"]
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
        if not (k.startswith("wsi::") or "wsi::" in k):
            continue
        nf, vf = f.get(k, [0, 0.0])
        nw, vw = w.get(k, [0, 0.0])
        n = max(nf, nw, 1)
        fm = 2.0 * vf * 1024 / n / 1e6
        wm = vw * 1024 / n / 1e6
        lines.append(f"\"{k}\",{n},{fm:.1f},{wm:.1f},{fm + wm:.1f}")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
