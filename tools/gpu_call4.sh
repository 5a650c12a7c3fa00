#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c4_gputests.log 2>&1; echo "gputests rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/c4_gputests.log | tail -15
for i in 1 2; do python tools/bench_dense.py --workload cfg003; python tools/bench_dense.py --workload perclass; done 2>&1 | grep -o '"ms_per_step": [0-9.]*'
bash tools/prof_variant.sh cfg003_onecall "--workload cfg003"
bash tools/prof_variant.sh cfg003_nopipe "--workload cfg003" APA_GEMM_WIDE_PIPE=0
bash tools/prof_variant.sh perclass_new "--workload perclass"
bash tools/prof_variant.sh perclass_nowidemid "--workload perclass" APA_GEMM_WIDE_MID=0
bash tools/prof_variant.sh perclass393 "--workload perclass --classes 393"
