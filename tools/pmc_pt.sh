#!/bin/bash
# SQ counters of the persistent-tile kernel on the batch-1024 linear1 product: tools/pmc_pt.sh <outdir>
OUT=$(realpath -m $1); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  VCT_GEMM_PT=3 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/p_$tag -o x -- python $R/tools/pt_big_loop.py > $OUT/p_$tag.log 2>&1
done
cd $R; python tools/pmc_summary.py $OUT gemm_pt
