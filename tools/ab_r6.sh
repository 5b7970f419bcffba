#!/bin/bash
# round 6 same-box A/B of the timed step: tools/ab_r6.sh "ENV=..." "ENV2=..."   (three alternating rounds)
run() { tag="$1"; env $1 python bench.py --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line --steps 80 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); g=j['roofline'].get('generator_gemms',{}); print('$tag', j['value'], j['ms_per_step'], {k:v['ms'] for k,v in g.items()})"; }
for rep in 1 2 3; do for s in "$@"; do run "$s"; done; done
