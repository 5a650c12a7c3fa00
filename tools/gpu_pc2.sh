#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 600 python -m pytest tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py tests/test_random_shapes_gpu.py -q -x -k "per_class or perclass" 2>&1 | grep -E "passed|failed|Error|error" | tail -4 | cut -c1-300
timeout 300 python tools/fuzz_all.py 150 5 perclass 2>&1 | grep -E "^FAIL|cases,"
for ps in 0 1 4 8; do echo "psplit=$ps"; APA_PC_ACT_PSPLIT=$ps python tools/bench_dense.py --workload perclass 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; APA_PC_ACT_PSPLIT=$ps bash tools/prof_dense.sh pcact --workload perclass 2>&1 | grep -E "pc_bwd_act|m1_colsum"; done
