#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "whole_step or cfg003 or benchmark_shape or wide_gemm or pose_head" > gpurun_out/c7_tests.log 2>&1; echo "tests rc=$?"
grep -E "^FAILED|passed|failed|Error" gpurun_out/c7_tests.log | tail -8
for i in 1 2 3; do python tools/bench_dense.py --workload cfg003; done 2>&1 | grep -o '"ms_per_step": [0-9.]*'
bash tools/prof_variant.sh cfg003_nodx "--workload cfg003"
bash tools/prof_variant.sh cfg003_dx "--workload cfg003" APA_POSE_STEP_NODX=0
