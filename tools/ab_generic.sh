# usage: tools/ab_generic.sh "ENV=..." "ENV2=..."   -- runs bench twice per setting, prints value / ms / brackets
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-decode --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$tag', j['value'], j['ms_per_step'], j['roofline']['all_ms'])"; }
for rep in 1 2; do for s in "$@"; do run "$s" $s; done; done
