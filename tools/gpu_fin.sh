#!/bin/bash
# A/B of the finalize / colsum block shapes (interleaved, same box)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 900 python -m pytest tests/test_attn_pool_gpu.py tests/test_head_gpu.py tests/test_dense_gpu.py -q -x 2>&1 | tail -3 | cut -c1-300
run() { label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-14s step %.2f us  fwd %.2f  bwd %.2f (frac %.3f)' % ('$label', d['ms_per_step']*1e3, d['roofline_fwd']['kernel_avg_us'], d['roofline']['kernel_avg_us'], d['roofline']['frac']))"
}
for i in 1 2 3 4; do
  run new A=1
  run old APA_M1_FIN_CW=256 APA_M1_COLSUM_COLS=32
  run fin-only APA_M1_COLSUM_COLS=32
  run col-only APA_M1_FIN_CW=256
  run col8 APA_M1_COLSUM_COLS=8
done
for i in 1 2 3; do
python tools/bench_dense.py --workload eval002 | grep -o '"images_per_sec": [0-9.]*, "ms_per_step": [0-9.]*'
APA_M1_FIN_CW=256 python tools/bench_dense.py --workload eval002 | grep -o '"images_per_sec": [0-9.]*, "ms_per_step": [0-9.]*'
done
