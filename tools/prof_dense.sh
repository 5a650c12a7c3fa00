#!/bin/bash
# rocprofv3 kernel-trace stats of one tools/bench_dense.py workload: prof_dense.sh <tag> <workload args...>
tag=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/prof_$tag; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/bench_dense.py "$@" > $O/bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - <<PY
import csv,glob
f=glob.glob('$O/stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(open('$O/bench.log').read().strip().splitlines()[-1][:200])
for r in rows[:22]:
    n=r['Name'].replace('void apa::','').replace('apa::','').replace('(anonymous namespace)::','').split('(')[0]
    print('%-70s calls %5s avg %8.2f us  %5s%%'%(n[:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
