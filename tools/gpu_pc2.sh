#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 600 python -m pytest tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py tests/test_random_shapes_gpu.py tests/test_head_gpu.py -q -x -k "per_class or perclass or step" 2>&1 | grep -E "passed|failed|Error|error" | tail -4 | cut -c1-300
timeout 300 python tools/fuzz_all.py 100 6 step,perclass 2>&1 | grep -E "^FAIL|cases,"
for i in 1 2; do python tools/bench_dense.py --workload perclass --classes 393 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
