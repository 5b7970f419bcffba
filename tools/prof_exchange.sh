#!/bin/bash
# rocprofv3 kernel stats of the exchange path at world size 1 (bench.py --force-exchange): tools/prof_exchange.sh <outdir> [sharded|allreduce]
OUT=$(realpath -m $1); kind=${2:-sharded}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_ex_$kind -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line --force-exchange --exchange $kind > $OUT/bench_$kind.json 2> $OUT/bench_$kind.err
db=$(find /tmp/prof_ex_$kind -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db > $OUT/kernel_stats_$kind.txt 2>&1
