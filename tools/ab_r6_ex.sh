#!/bin/bash
# same-box A/B of the exchange path at world size 1: tools/ab_r6_ex.sh "ENV=..." "ENV2=..."
run() { tag="$1"; env $1 python bench.py --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line --force-exchange --exchange sharded --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); g=j['roofline'].get('generator_gemms',{}); print('$tag', j['value'], j['ms_per_step'], {k:v['ms'] for k,v in g.items()})"; }
for rep in 1 2 3; do for s in "$@"; do run "$s"; done; done
