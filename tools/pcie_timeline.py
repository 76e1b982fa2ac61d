#!/usr/bin/env python
"""Kernels and memory copies of the LAST loader-fed steps out of a `rocprofv3 --kernel-trace --memory-copy-trace` rocpd database, in start order, with the
queue / stream each ran on: what runs beside the H2D feature copies of the pinned-host loader.  `python tools/pcie_timeline.py x.db [out.txt] [n_rows]`"""
import sqlite3
import sys


def main(path, out=None, n_rows=400):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    lines = []
    kc = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    mc_name = "memory_copies" if "memory_copies" in tabs else next((t for t in tabs if "memory_cop" in t), None)
    mc = [r[1] for r in cur.execute(f"pragma table_info({mc_name})")] if mc_name else []
    lines.append("# kernels columns: " + ",".join(kc))
    lines.append(f"# {mc_name} columns: " + ",".join(mc))
    namecol = "name" if "name" in kc else "kernel_name"
    qcol = "queue_id" if "queue_id" in kc else ("queue" if "queue" in kc else None)
    scol = "stream_id" if "stream_id" in kc else ("stream" if "stream" in kc else None)
    rows = [(s, e, "K", n, q, st) for n, s, e, q, st in cur.execute(
        f"select {namecol}, start, end, {qcol or 'null'}, {scol or 'null'} from kernels")]
    if mc_name:
        size = "size" if "size" in mc else ("bytes" if "bytes" in mc else "null")
        nm = "name" if "name" in mc else "null"
        sc = "stream_id" if "stream_id" in mc else ("stream" if "stream" in mc else "null")
        rows += [(s, e, "C", f"{n} {b}", None, st) for n, s, e, b, st in cur.execute(f"select {nm}, start, end, {size}, {sc} from {mc_name}")]
    rows.sort()
    rows = rows[-int(n_rows):]
    t0 = rows[0][0]
    for s, e, k, n, q, st in rows:
        lines.append(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {k} q={q} s={st}  {str(n)[:110]}")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    else:
        print(text)


def summary(path, out=None):
    """Per step (optimizer launch to optimizer launch): wall time, kernel time on the caller's stream, the H2D feature copies inside it; then the
    kernels whose time differs most between a resident-batch step and a loader-fed pinned-host step."""
    import collections
    import re
    cur = sqlite3.connect(path).cursor()
    rows = [(s, e - s, "K", st, n) for n, s, e, st in cur.execute("select name, start, end, stream_id from kernels")]
    rows += [(s, e - s, "C", st, f"{n} {b}") for n, s, e, b, st in cur.execute("select name, start, end, size, stream_id from memory_copies")]
    rows.sort()
    main_stream = collections.Counter(r[3] for r in rows if r[2] == "K").most_common(1)[0][0]
    adam = [i for i, r in enumerate(rows) if "adam_step" in r[4]]
    lines = ["# step: wall us | kernel time on the caller's stream us | H2D feature copies (>= 8 MB): count, busy us, first start, last end (us after the previous optimizer launch)"]
    plain, fed = None, None
    for a, b in zip(adam[:-1], adam[1:]):
        seg = rows[a + 1:b + 1]
        wall = (rows[b][0] + rows[b][1] - rows[a][0] - rows[a][1]) / 1e3
        if wall > 5e4:
            continue                                       # a set-up phase between the legs
        cp = [r for r in seg if r[2] == "C" and "HOST_TO_DEVICE" in r[4] and int(r[4].split()[-1]) >= 8000000]
        busy = sum(r[1] for r in seg if r[2] == "K" and r[3] == main_stream) / 1e3
        lines.append(f"{wall:9.1f} {busy:9.1f} {len(cp):4d} {sum(r[1] for r in cp) / 1e3:9.1f} "
                     f"{(cp[0][0] - rows[a][0]) / 1e3 if cp else -1:9.1f} {(cp[-1][0] + cp[-1][1] - rows[a][0]) / 1e3 if cp else -1:9.1f}")
        if not cp and plain is None and wall < 6.5e3:
            plain = (a, b)
        if len(cp) == 24 and fed is None:
            fed = (a, b)

    def agg(a, b):
        d = collections.OrderedDict()
        for r in rows[a + 1:b + 1]:
            if r[2] == "K":
                n = re.sub(r"^void ", "", re.sub(r"\(.*", "", r[4]))[:70] + f" [stream {r[3]}]"
                d.setdefault(n, [0, 0.0])
                d[n][0] += 1
                d[n][1] += r[1] / 1e3
        return d
    if plain and fed:
        A, B = agg(*plain), agg(*fed)
        lines.append("# kernel: launches, total us in a resident-batch step | launches, total us in a loader-fed pinned-host step (differences > 40 us)")
        for k in sorted(set(A) | set(B), key=lambda k: -abs(A.get(k, [0, 0])[1] - B.get(k, [0, 0])[1])):
            x, y = A.get(k, [0, 0.0]), B.get(k, [0, 0.0])
            if abs(x[1] - y[1]) > 40:
                lines.append(f"{k:95s} {x[0]:3d} {x[1]:8.1f} | {y[0]:3d} {y[1]:8.1f}")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    else:
        print(text)


if __name__ == "__main__":
    if sys.argv[1] == "--summary":
        summary(*sys.argv[2:4])
    else:
        main(*sys.argv[1:4])
