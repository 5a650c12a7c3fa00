#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 600 python -m pytest tests/test_head_gpu.py tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py tests/test_random_shapes_gpu.py -q -x -k "loss or cfg003 or golden or gen_losses" 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2 3; do python tools/bench_dense.py --workload cfg003 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
bash tools/prof_dense.sh c3l2 --workload cfg003 2>&1 | grep -E "pose_l2|sum_scale"
