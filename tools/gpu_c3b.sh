#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 600 python -m pytest tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
run() { label=$1; shift; env "$@" python tools/bench_dense.py --workload cfg003 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$label /"; }
for i in 1 2 3; do
  run new A=1
  run old APA_GEMM_SPLITS=3
done
