# usage: tools/ab_flags.sh "<bench flags A>" "<bench flags B>" ...   -- runs bench twice per flag set
run() { python bench.py --no-cpu-baseline --no-decode --steps 40 --warmup 10 $1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('[$1]', j['value'], j['ms_per_step'], j['north_star']['ms'])"; }
for rep in 1 2; do for s in "$@"; do run "$s"; done; done
