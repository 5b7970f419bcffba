import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall() if "grid_x" in cols else None
if rows is None:
    print(cols); sys.exit()
agg = collections.defaultdict(list)
for n, s, e, g in rows:
    if "decode_gemv" in n or "argmax" in n:
        key = (n.split("(")[0].replace("vct::", "").replace("void ", "")[:60], g)
        agg[key].append((e - s) / 1e3)
for k, v in sorted(agg.items()):
    v2 = sorted(v)
    print(f"{k[0]:62s} grid {k[1]:7d} calls {len(v):5d} med {v2[len(v2)//2]:7.2f} us min {v2[0]:7.2f}")
