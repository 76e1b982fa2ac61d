import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, start, end from regions").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for n, s, e in rows:
    d = (e - s) / 1e6; a = agg[n]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])[:14]:
    print(f"{k:44s} calls {v[0]:7d} total {v[1]:9.1f} ms  max {v[2]:8.2f} ms")
slow = sorted([(e - s) / 1e6, n, s] for n, s, e in rows if (e - s) / 1e6 > 20)
print("calls > 20 ms:", [(round(d, 1), n) for d, n, s in slow[-25:]])
