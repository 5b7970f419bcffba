#!/bin/bash
# ONE evidence pass on the GPU box (through gpurun), final code of the round: everything profiles/rNN_* is assembled from
# (tools/collect_evidence.py).  usage: bash tools/evidence_pass.sh <dir under gpurun_out> <tag of the eager step profile>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; TAG=$2; mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_list
rocprofv3 --kernel-trace --stats -d /tmp/prof_list -o r -- python $R/bench.py --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cd $R
db=$(find /tmp/prof_list -name "*results.db" | head -1)
python tools/rocpd_summary.py $db --skip-frac 0.35 > $OUT/bench_list_kernel_summary.txt 2>&1
bash tools/prof_step.sh $TAG
bash tools/pmc_bench.sh $OUT/pmc > /dev/null 2>&1
python tools/pmc_bench_summary.py $OUT/pmc > $OUT/pmc_per_kernel.txt 2>&1
rm -rf $OUT/pmc/bench_*/x_counter_collection.csv $OUT/pmc/bench_*/x_kernel_trace.csv 2>/dev/null
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -4 > $OUT/gpu_tests.txt
tail -2 $OUT/gpu_tests.txt; tail -c 400 $OUT/bench_n1.json
