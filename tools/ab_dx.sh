run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-decode --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$tag', j['value'], j['ms_per_step'], j['roofline']['all_ms'])"; }
run nt X=1
run nn VCT_GEMM256=1
run nt X=1
run nn VCT_GEMM256=1
