#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel table:
calls, total / average / min / max duration, share of GPU time.  Optionally skip the first N
dispatches of every kernel (warm-up).   python tools/rocpd_summary.py results.db [--skip-frac 0.25]"""
import argparse
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("vct::", "").replace("void ", "")
    name = re.sub(r"unsigned short", "bf16", name)
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip-frac", type=float, default=0.0, help="drop this leading fraction of the timeline (warm-up)")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        print("no kernel dispatches"); return
    t0, t1 = rows[0][1], rows[-1][2]
    cut = t0 + (t1 - t0) * a.skip_frac
    agg = {}
    for n, s, e in rows:
        if s < cut:
            continue
        d = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d[0] += 1; d[1] += e - s; d[2] = min(d[2], e - s); d[3] = max(d[3], e - s)
    tot = sum(v[1] for v in agg.values())
    span = t1 - cut
    print(f"# {a.db}: {sum(v[0] for v in agg.values())} dispatches, kernel time {tot/1e6:.3f} ms over a {span/1e6:.3f} ms window "
          f"(GPU busy {100*tot/span:.1f}%)")
    print(f"{'kernel':110s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:110s} {v[0]:6d} {v[1]/1e6:10.3f} {v[1]/v[0]/1e3:10.2f} {v[2]/1e3:9.2f} {v[3]/1e3:9.2f} {100*v[1]/tot:6.2f}")


if __name__ == "__main__":
    main()
