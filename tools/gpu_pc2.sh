#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 600 python -m pytest tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py -q -x -k "per_class" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 | cut -c1-300
for i in 1 2; do
  python tools/bench_dense.py --workload perclass 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  APA_PC_ZT_DMA=0 python tools/bench_dense.py --workload perclass 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
bash tools/prof_dense.sh pcdma --workload perclass 2>&1 | head -14
