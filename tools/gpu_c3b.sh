#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 600 python -m pytest tests/test_dense_gpu.py -q -x -k "cfg003" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 | cut -c1-300
for i in 1 2 3; do
  python tools/bench_dense.py --workload cfg003 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
python tools/exp_host_dense.py 2>&1 | grep -E "host enq"
