#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
for i in 1 2 3 4; do
  APA_POSE_RPB=32 python tools/bench_dense.py --workload cfg003 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/rpb32 /'
  APA_POSE_RPB=16 python tools/bench_dense.py --workload cfg003 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/rpb16 /'
done
