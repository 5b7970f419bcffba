#!/usr/bin/env python
"""Per-kernel PMC summary of a bench run: for each kernel name (shortened) the per-dispatch mean of every
counter and, for the largest-duration dispatch class of each kernel (e.g. the generator GEMM among the NT
GEMMs), the values of that class.   python tools/pmc_bench_summary.py gpurun_out/pmc2"""
import collections, csv, glob, os, re, sys
root = sys.argv[1]

def short(n):
    n = re.sub(r"\(.*\)$", "", n).replace("vct::", "").replace("void ", "").replace("unsigned short", "bf16")
    return n[:70]

dur = {}   # (kernel, dispatch_id) -> us from the same pass
out = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, "bench_*"))):
    f = os.path.join(d, "x_counter_collection.csv")
    if not os.path.isfile(f):
        continue
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> sum
    meta = {}
    for r in rows:
        key = (short(r["Kernel_Name"]), r["Dispatch_Id"])
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[key] = (int(r["Grid_Size"]) if "Grid_Size" in r else 0)
    kt = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
          for r in csv.DictReader(open(os.path.join(d, "x_kernel_trace.csv")))}
    groups = collections.defaultdict(list)
    for (k, disp), c in per.items():
        groups[(k, meta[(k, disp)])].append((kt.get(disp, 0.0), c))
    for (k, grid), lst in groups.items():
        n = len(lst)
        avg_us = sum(t for t, _ in lst) / n
        ent = out[(k, grid)]
        ent["calls"] = n; ent["avg_us"] = avg_us
        for cname in lst[0][1]:
            ent[cname] = sum(c[cname] for _, c in lst) / n
print(f"{'kernel (grid size)':86s} {'calls':>5s} {'avg_us':>8s}  counters (mean per dispatch)")
for (k, grid), ent in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"])[:28]:
    cs = " ".join(f"{c}={v:.4g}" for c, v in ent.items() if c not in ("calls", "avg_us"))
    print(f"{(k + ' [' + str(grid) + ']'):86s} {ent['calls']:5d} {ent['avg_us']:8.1f}  {cs}")
