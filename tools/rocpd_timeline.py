#!/usr/bin/env python
"""One training step of a rocprofv3 --kernel-trace database as a timeline: for every kernel its queue (stream), start
offset, duration, and the idle gap since the previous kernel on the same queue.  The step is the window between two
consecutive bump_step kernels in the MIDDLE of the trace (argv[2] = fraction of the trace, default 0.45): bench.py's timed steps --
the last steps of a bench.py trace are its second pass with EVERY timing bracket on, and each bracket (two event records) shows
up as ~6 us of idle time in front of the kernel behind it (timelines up to round 4 / r5a..r5p were taken there).
Dev tool.   python tools/rocpd_timeline.py results.db [fraction]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(.*\)$", "", n).replace("vct::", "").replace("void ", "")
    n = re.sub(r"unsigned short", "bf16", n)
    return n[:64]


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else ("stream_id" if "stream_id" in cols else None))
    rows = c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
    bumps = [i for i, r in enumerate(rows) if "bump_step" in r[0]]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
    k = max(1, min(len(bumps) - 2, int(len(bumps) * frac)))
    a, b = bumps[k - 1] + 1, bumps[k] + 1        # one full step (profiling overhead included)
    step = rows[a:b]
    t0 = step[0][1]
    last_end = {}
    print(f"# step of {len(step)} kernels, {(step[-1][2] - t0) / 1e3:.1f} us wall (profiled), queue column: {qcol}")
    busy = sum(e - s for _n, s, e, _q in step)
    print(f"# sum of kernel durations {busy / 1e3:.1f} us")
    for n, s, e, q in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f"q{q!s:>3} +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {short(n)}")


if __name__ == "__main__":
    main()
