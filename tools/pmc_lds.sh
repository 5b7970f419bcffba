#!/bin/bash
# LDS bank-conflict counters of the vocabulary projection (run on the GPU box): tools/pmc_lds.sh <outdir>
OUT=$(realpath -m $1); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/p_$tag -o x -- python $R/tools/gen_fwd_bench.py > $OUT/p_$tag.log 2>&1
  python - $OUT/p_$tag/x_counter_collection.csv <<'PY'
import csv, sys, collections
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counter file", e); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "gemm" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
  tail -2 $OUT/p_$tag.log | cut -c1-200
done
