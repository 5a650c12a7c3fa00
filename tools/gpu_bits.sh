#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 900 python -m pytest tests/test_attn_pool_gpu.py tests/test_head_gpu.py -q -x 2>&1 | tail -3 | cut -c1-200
run() { label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-14s step %.2f us  fwd %.2f  bwd %.2f (frac %.3f)' % ('$label', d['ms_per_step']*1e3, d['roofline_fwd']['kernel_avg_us'], d['roofline']['kernel_avg_us'], d['roofline']['frac']))"
}
run f32 A=1; run f32 A=1
EXTRA="--batch 512 --steps 30" run f32_n512 A=1
EXTRA="--dtype bf16" run bits_bf16 A=1; EXTRA="--dtype bf16" run hash_bf16 APA_M1_KEEP_BITS=0
EXTRA="--dtype bf16 --batch 512 --steps 30" run bits_bf16_512 A=1; EXTRA="--dtype bf16 --batch 512 --steps 30" run hash_bf16_512 APA_M1_KEEP_BITS=0
