#!/bin/bash
# PMC passes over the three vocabulary products in the library's own plan (gemm256_kernel NT, g32_kernel NN / TN): LDS and MFMA pipe
# occupancy, bank conflicts, waits.  Run on the GPU box: tools/pmc_vocab.sh <outdir>
OUT=$(realpath -m $1); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export VCT_BENCH_WARM=5 VCT_BENCH_ITERS=6
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/p_$tag -o x -- python $R/tools/gemm_bench.py --plan gen_fwd gen_dx gen_dw > $OUT/p_$tag.log 2>&1
done
python - $OUT <<'PY'
import csv, sys, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p_*/x_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        if "g32" in k or "gemm256" in k or "splitk" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    print("   ", {c: round(v) for c, v in sorted(m.items())})
    g = m.get("GRBM_GUI_ACTIVE", 0)
    if g:
        g = g / 8.0                      # GRBM_GUI_ACTIVE comes summed over the 8 XCDs; the SQ counters over all CUs / SIMDs
        cu, simd = 256, 1024
        print(f"    cycles per XCD {g:.0f}; MFMA pipe busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd / g:.3f} of SIMD-cycles; "
              f"LDS index active {m.get('SQ_LDS_IDX_ACTIVE', 0) / cu / g:.3f} of CU-cycles, bank-conflict cycles / active "
              f"{m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}; "
              f"waves waiting (any) {m.get('SQ_WAIT_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1):.3f} of wave-cycles, "
              f"waiting on LDS instr issue {m.get('SQ_WAIT_INST_LDS', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
