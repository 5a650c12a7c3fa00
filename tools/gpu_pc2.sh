#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 600 python -m pytest tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py tests/test_random_shapes_gpu.py -q -x -k "per_class or perclass" 2>&1 | grep -E "passed|failed|Error|error" | tail -4 | cut -c1-300
for i in 1 2 3; do python tools/bench_dense.py --workload perclass 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
bash tools/prof_dense.sh pcprep --workload perclass 2>&1 | grep -E "pc_prep"
