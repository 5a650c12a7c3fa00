#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -s -k "wide_gemm or whole_step_in_one_call or benchmark_shape or accumulate_gradients_divides or overlapped_micro or fused_input_relu" > gpurun_out/c2_newtests.log 2>&1; echo "newtests rc=$?"
grep -E "passed|failed|logits max abs|Error|assert" gpurun_out/c2_newtests.log | tail -40
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/c2_gputests.log 2>&1; echo "gputests rc=$?"
tail -8 gpurun_out/c2_gputests.log
for i in 1 2; do python tools/bench_dense.py --workload cfg003; python tools/bench_dense.py --workload cfg003 --per-op; done 2>&1 | grep -o '"ms_per_step": [0-9.]*'
bash tools/prof_variant.sh cfg003_onecall "--workload cfg003"
bash tools/prof_variant.sh cfg003_onecall_nopipe "--workload cfg003" APA_GEMM_WIDE_PIPE=0
bash tools/prof_variant.sh cfg003_onecall_nowide "--workload cfg003" APA_GEMM_WIDE=0
bash tools/prof_variant.sh cfg003_perop "--workload cfg003 --per-op" APA_GEMM_WIDE=0
