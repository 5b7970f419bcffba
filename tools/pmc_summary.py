#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output dirs: per (run, kernel-substring) mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
for d in sorted(glob.glob(os.path.join(root, "*"))):
    f = os.path.join(d, "x_counter_collection.csv")
    if not os.path.isfile(f): continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    kt = os.path.join(d, "x_kernel_trace.csv")
    dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if filt in r["Kernel_Name"]]
    print(os.path.basename(d), f"dispatches={len(dur)} avg_us={sum(dur)/max(len(dur),1):.1f}")
    for k, v in acc.items():
        # counters are reported per dispatch (possibly per-dimension rows): sum rows per dispatch
        n = max(len(dur), 1)
        print(f"   {k:32s} {sum(v)/n:16.0f}")
