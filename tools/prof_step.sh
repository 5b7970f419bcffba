#!/bin/bash
# rocprofv3 kernel-trace summary of the bench step (run on the GPU box through gpurun).  usage: tools/prof_step.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line --executor eager "$@" > /tmp/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/prof_$tag -name "*results.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_summary.py $db > gpurun_out/${tag}_kernel_summary.txt 2>&1; python tools/rocpd_timeline.py $db > gpurun_out/${tag}_timeline.txt 2>&1; else ls -R /tmp/prof_$tag | head -30; tail -5 /tmp/prof_$tag.log; fi
grep -E "^\{|Error|error|Traceback" /tmp/prof_$tag.log | cut -c1-300 | head -5
