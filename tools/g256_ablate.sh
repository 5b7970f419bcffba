#!/bin/bash
# Ablation of the persistent 256x256 kernel per operand layout (run on the GPU box): tools/g256_ablate.sh <outdir>
# VCT_GEMM256_DBG: 1 = no MFMA work, 2 = no operand DMA after the first stage, 4 = no epilogue
OUT=$(realpath -m $1); mkdir -p $OUT; cd $GRAFT_REPO_ROOT/tools
for shape in gen_fwd gen_dx gen_dw; do
  for dbg in 0 1 2 4 6 5; do
    echo -n "dbg=$dbg " >> $OUT/ablate.txt
    VCT_GEMM256=15 VCT_GEMM256_DBG=$dbg python gemm_one.py $shape 0 0 20 >> $OUT/ablate.txt 2>&1
  done
done
cd /tmp; export TMPDIR=/tmp
for shape in gen_dx gen_dw; do
  for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$shape -o x -- env VCT_GEMM256=15 python $GRAFT_REPO_ROOT/tools/gemm_one.py $shape 0 0 3 > $OUT/pmc_$shape.log 2>&1
    python - $OUT/pmc_$shape/x_counter_collection.csv >> $OUT/ablate.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "gemm" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  done
done
