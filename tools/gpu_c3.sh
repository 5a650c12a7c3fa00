#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py tests/test_head_gpu.py -q -x -k "pose or cfg003 or dense or bf16" 2>&1 | tail -4
python tools/bench_dense.py --workload cfg003
APA_POSE_PL_FAST=0 python tools/bench_dense.py --workload cfg003
bash tools/prof_dense.sh r02_cfg003 --workload cfg003 2>&1 | grep "kernel\|images" | head -24
