#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 600 python -m pytest tests/test_bf16_parity_gpu.py -q -k "per_class" 2>&1 | tail -3
bash tools/prof_dense.sh r02_perclass --workload perclass 2>&1 | grep "kernel\|images" | head -11
