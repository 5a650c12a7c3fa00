#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bf16_parity_gpu.py -q -s > gpurun_out/pytest_b_bf16.log 2>&1; echo "bf16 rc=$?"
tail -30 gpurun_out/pytest_b_bf16.log
timeout 1500 python -m pytest tests -m gpu -q -k "per_class or bench_prints" > gpurun_out/pytest_b.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_b.log
timeout 300 python bench.py --no-extra --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_b.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us'], d['roofline_fwd']['kernel_avg_us'])"
python tools/bench_dense.py --workload perclass
APA_PC_FUSED=0 python tools/bench_dense.py --workload perclass
APA_PC_BM=64 python tools/bench_dense.py --workload perclass
APA_PC_DW_SPLITS=8 python tools/bench_dense.py --workload perclass
APA_PC_DW_SPLITS=32 python tools/bench_dense.py --workload perclass
