#!/bin/bash
# round-4 call 1: new tests, end-to-end lines, e2e profile
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s -k "benchmark_shape or fused_input_relu or accumulate_gradients_divides or two_ranks" > gpurun_out/c1_newtests.log 2>&1; echo "newtests rc=$?" 
tail -5 gpurun_out/c1_newtests.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1_gputests.log 2>&1; echo "gputests rc=$?"
tail -3 gpurun_out/c1_gputests.log
timeout 900 bash tools/profile_e2e.sh r04 > gpurun_out/c1_e2e.log 2>&1; echo "e2e rc=$?"
tail -60 gpurun_out/c1_e2e.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra-only cfg002_eval_e2e,cfg003_train_e2e > gpurun_out/c1_bench_e2e.json 2> gpurun_out/c1_bench_e2e.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/c1_bench_e2e.json').read().strip().splitlines()[-1]); print(json.dumps(d['extra'], indent=1)[:3000])"
