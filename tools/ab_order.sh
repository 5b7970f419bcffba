run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-decode --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$tag', j['value'], j['ms_per_step'], j['roofline']['all_ms'])"; }
run new X=1
run old VCT_GEMM256_ORDER=0 VCT_GEMM_ORDER=0
run allshort VCT_GEMM_ORDER=1
run new X=1
run old VCT_GEMM256_ORDER=0 VCT_GEMM_ORDER=0
