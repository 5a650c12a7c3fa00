#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "per_class or perclass or pc_ or whole_step or cfg003 or hmdb" > gpurun_out/c5_tests.log 2>&1; echo "tests rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/c5_tests.log | tail -8
for i in 1 2; do python tools/bench_dense.py --workload cfg003; python tools/bench_dense.py --workload perclass; done 2>&1 | grep -o '"ms_per_step": [0-9.]*'
bash tools/prof_variant.sh cfg003_onecall "--workload cfg003"
bash tools/prof_variant.sh perclass_ct64 "--workload perclass"
bash tools/prof_variant.sh perclass_ct128 "--workload perclass" APA_PC_DW_CT=128
