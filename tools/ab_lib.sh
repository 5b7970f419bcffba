run() { tag="$1"; shift; env "$@" python bench.py --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line --steps 80 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$tag', j['value'], j['ms_per_step'], j['roofline']['all_ms']['adam'])"; }
for rep in 1 2 3; do
run new X=1
run old VCT_LIB_PATH=$PWD/video-captioning-transformer_amd/libvct_hip_ab.so
done
