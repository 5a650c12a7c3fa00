#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pc393; rm -rf $O; mkdir -p $O; cd $R
for k in 393 51; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$k -- python tools/bench_dense.py --workload perclass --classes $k > $O/k$k.log 2>&1
f=$(find $O/k$k -name '*kernel_stats.csv' | head -1)
echo "== K=$k"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls'])>=20 and float(r['Percentage'])>1: print('%-86s %5s %9.2f %6s'%(r['Name'].replace('(anonymous namespace)::','').split('(')[0][:86], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
