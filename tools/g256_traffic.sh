#!/bin/bash
# Fabric traffic of the persistent 256x256 vocabulary projection under cache-policy / order variants (run on the GPU box):
#   tools/g256_traffic.sh <outdir>      FETCH_SIZE / WRITE_SIZE / TCC hit-miss passes over tools/gen_fwd_bench.py per variant + plain timing
set -u
OUT=$(realpath -m $1); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
variants=(${VARIANTS:-"base" "nt0:VCT_GEMM_NT=0" "a_nt:VCT_GEMM256_DBG=8" "b_nt:VCT_GEMM256_DBG=16" "order0:VCT_GEMM256_ORDER=0"})
for v in "${variants[@]}"; do
  name=${v%%:*}; envs=""; [[ "$v" == *:* ]] && envs=${v#*:}
  echo "== $name ($envs)" >> $OUT/timing.txt
  env $envs python $R/tools/gen_fwd_bench.py >> $OUT/timing.txt 2>&1
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ"; do
    tag=$(echo $pass | cut -d' ' -f1)
    env $envs rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/${name}_$tag -o x -- python $R/tools/gen_fwd_bench.py > $OUT/${name}_$tag.log 2>&1
  done
done
python $R/tools/pmc_summary.py $OUT gemm256 > $OUT/summary.txt 2>&1
find $OUT -name "x_counter_collection.csv" -delete; find $OUT -name "x_kernel_trace.csv" -delete
cat $OUT/timing.txt | grep -E "==|bias \+" ; cat $OUT/summary.txt
