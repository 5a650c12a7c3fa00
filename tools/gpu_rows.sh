#!/bin/bash
# pose-head backward: one-pass rows kernel vs the round-1 three-launch form
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_bf16_parity_gpu.py tests/test_head_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 | cut -c1-300
for i in 1 2 3; do
  python tools/bench_dense.py --workload cfg003 2>/dev/null | grep -o '"images_per_sec": [0-9.]*, "ms_per_step": [0-9.]*'
  APA_POSE_BWD_ROWS=0 python tools/bench_dense.py --workload cfg003 2>/dev/null | grep -o '"images_per_sec": [0-9.]*, "ms_per_step": [0-9.]*'
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_rows -o rows -- python $R/tools/bench_dense.py --workload cfg003 > /dev/null 2>&1
python - <<'PY'
import csv,glob,os
f=sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/prof_rows/**/*kernel_stats.csv',recursive=True),key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:24]:
    print('%-90s %6s %9.2f %6s'%(r['Name'][:90],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
PY
