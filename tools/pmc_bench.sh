#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (separate passes, gfx950 rules): run on the GPU box
set -u
OUT=$(realpath -m $1); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/bench_$tag -o x -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode --executor eager --no-overlap-dw --no-b1024 --no-other-configs --no-exchange-line > $OUT/bench_$tag.log 2>&1
done
