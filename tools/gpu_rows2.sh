#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 | cut -c1-300
for cfg in "0 16" "0 32" "1 16" "2 16" "4 16" "7 16"; do set -- $cfg
echo "dbg=$1 rpb=$2"; APA_POSE_DBG=$1 APA_POSE_RPB=$2 bash tools/prof_dense.sh rowsd --workload cfg003 2>&1 | grep -E "pose_bwd_rows|m1_colsum" | cut -c1-160
done
