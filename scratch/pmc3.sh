#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/pmc3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc3 -- python $R/tools/bench_dense.py --steps 10 --warmup 2 > /dev/null 2>&1
f=$(find $R/gpurun_out/pmc3 -name "*counter_collection.csv"|head -1)
python - "$f" <<'PY'
import csv,sys
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if 'gemm_bf16' in n or 'dppre' in n: acc[n[:110]][r['Counter_Name']].append(float(r['Counter_Value']))
for n,d in acc.items():
    print(n)
    print('   ', {k: round(sum(v)/len(v)) for k,v in d.items()})
PY
