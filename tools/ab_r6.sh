#!/bin/bash
# round 6 same-box A/B of the timed step: tools/ab_r6.sh  (libvct_hip_ab.so = tools/ab_build.sh HEAD vct_gemm_bf16_kernel.h)
run() { tag="$1"; shift; env "$@" python bench.py --no-cpu-baseline --no-decode --no-b1024 --no-other-configs --no-exchange-line --steps 80 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); g=j['roofline'].get('generator_gemms',{}); print('$tag', j['value'], j['ms_per_step'], {k:v['ms'] for k,v in g.items()})"; }
for rep in 1 2 3; do
run "new(g32 NT + balanced db)" X=1
run "new, VCT_GEMM32=0" VCT_GEMM32=0
run "old general kernel" VCT_LIB_PATH=$PWD/video-captioning-transformer_amd/libvct_hip_ab.so
done
