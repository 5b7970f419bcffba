#!/bin/bash
# rocprofv3 kernel-trace of the bench step under the LIST executor (what bench.py times): timeline of one step.  usage: tools/prof_step_list.sh <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line "$@" > /tmp/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/prof_$tag -name "*results.db" | head -1)
python tools/rocpd_timeline.py $db > gpurun_out/${tag}_timeline.txt 2>&1
grep -E "^\{|Error|error|Traceback" /tmp/prof_$tag.log | cut -c1-200 | head -3
