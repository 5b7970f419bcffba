# usage: tools/ab_env.sh "ENV=..." "ENV2=..."   -- short bench runs (timed region only), twice per setting, same box
run() { tag="$1"; env $1 python bench.py --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line --steps 80 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$tag', j['value'], j['ms_per_step'])"; }
for rep in 1 2; do for s in "$@"; do run "$s"; done; done
