#!/usr/bin/env python
"""Per-kernel L2 (TCC) hit rate from one rocprofv3 PMC pass: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), plus request counts
(MI355X_MICROARCH.md §L2).  Usage: pmc_l2.py <counter_collection.csv> [out.csv]"""
import collections
import csv
import sys


def main(path, out=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            launches[name] += 1
    cols = sorted({c for v in agg.values() for c in v})
    lines = ["kernel,launches," + ",".join(f"{c}_per_launch" for c in cols) + ",l2_hit_rate"]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        if "wsi::" not in k:
            continue
        n = max(launches[k], 1)
        hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
        rate = hit / (hit + miss) if hit + miss > 0 else float("nan")
        lines.append(f"\"{k}\",{n}," + ",".join(f"{v.get(c, 0.0) / n:.0f}" for c in cols) + f",{rate:.4f}")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
