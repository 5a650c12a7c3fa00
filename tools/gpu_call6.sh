#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c6_gputests.log 2>&1; echo "gputests rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/c6_gputests.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do python tools/bench_dense.py --workload cfg003; python tools/bench_dense.py --workload perclass; done 2>&1 | grep -o '"ms_per_step": [0-9.]*'
timeout 600 bash tools/fuzz_arms.sh > gpurun_out/c6_fuzz_arms.log 2>&1; tail -30 gpurun_out/c6_fuzz_arms.log
