#!/bin/bash
# PMC passes over single GEMM configs (run on the GPU box): tools/pmc_gemm.sh <outdir>
set -u
OUT=$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name shape tile nbuf
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ" "GRBM_GUI_ACTIVE FETCH_SIZE" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$1_$tag -o x -- python $R/tools/gemm_one.py "$2" $3 $4 3 > $OUT/$1_$tag.log 2>&1
  done
}
cd $R/tools
run gen_fwd "gen_fwd" 3 2
run gen_dw "gen_dw" 1 2
run ffn1_fwd "ffn1_fwd" 4 2
