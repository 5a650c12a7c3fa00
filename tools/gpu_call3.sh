#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c3_gputests.log 2>&1; echo "gputests rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/c3_gputests.log | tail -15
for i in 1 2; do python tools/bench_dense.py --workload cfg003; python tools/bench_dense.py --workload perclass; done 2>&1 | grep -o '"ms_per_step": [0-9.]*'
bash tools/prof_variant.sh cfg003_onecall "--workload cfg003"
bash tools/prof_variant.sh perclass_new "--workload perclass"
bash tools/prof_variant.sh perclass_old "--workload perclass" APA_PC_PREBITS=0 APA_PC_DW_SWZ=0 APA_PC_XENT_FOLD=0
bash tools/prof_variant.sh perclass_noprebits "--workload perclass" APA_PC_PREBITS=0
bash tools/prof_variant.sh perclass_noswz "--workload perclass" APA_PC_DW_SWZ=0
