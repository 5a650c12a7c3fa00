#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 600 python -m pytest tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
for w in 4 2 1; do echo "waves=$w"; APA_POSE_PL_WAVES=$w python tools/bench_dense.py --workload cfg003 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; APA_POSE_PL_WAVES=$w bash tools/prof_dense.sh c3pl --workload cfg003 2>&1 | grep -E "pose_pl"; done
