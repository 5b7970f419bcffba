#!/bin/bash
# kernel sequence of one batch-128 decode step (run on the GPU box).  usage: tools/decode_b128_prof.sh <tag> [B]
tag=$1; B=${2:-128}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dprof_$tag
rocprofv3 --kernel-trace -d /tmp/dprof_$tag -o r -- python $GRAFT_REPO_ROOT/tools/decode_b128_run.py $B > /tmp/dprof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/dprof_$tag -name "*results.db" | head -1)
python - "$db" > gpurun_out/${tag}_decode_seq.txt <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kv = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
cols = [r[1] for r in c.execute(f"pragma table_info({kv[0]})")]
rows = c.execute(f"select name, start, end, grid_x from {kv[0]} order by start").fetchall()
# last 26*3 kernels = the final three steps; print the last step's sequence and medians per (name, grid) over all
agg = collections.defaultdict(list)
for n, s, e, g in rows:
    agg[(n.split("(")[0].replace("vct::", "").replace("void ", "")[:80], g)].append((e - s) / 1e3)
tot = 0.0
print("per (kernel, grid): calls, median us")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v); print(f"{k[0]:82s} grid {k[1]:8d} calls {len(v):5d} med {v2[len(v2)//2]:7.2f}")
print("\nlast 30 kernels in order:")
for n, s, e, g in rows[-30:]:
    print(f"{(e-s)/1e3:7.2f} us  grid {g:8d}  {n.split('(')[0].replace('vct::','').replace('void ','')[:90]}")
PY
tail -3 /tmp/dprof_$tag.log
