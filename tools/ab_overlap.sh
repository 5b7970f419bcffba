run() { tag="$1"; shift; python bench.py --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$tag', j['value'], j['ms_per_step'])"; }
for rep in 1 2; do
run default
run no-dw --no-overlap-dw
run no-dw-kv --no-overlap-dw --no-overlap-kv
run no-enc --no-overlap-enc
run serial --no-overlap-dw --no-overlap-kv --no-overlap-enc
done
