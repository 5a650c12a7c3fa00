#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bigN; rm -rf $O; mkdir -p $O; cd $R
for B in 32 128 512; do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $B', d['value'], d['ms_per_step'], d['step_roofline_frac'])"
done
for B in 128 512; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$B -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --batch $B > $O/b$B.log 2>&1
f=$(find $O/b$B -name '*kernel_stats.csv' | head -1)
echo "== batch $B"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls'])>=50: print('%-60s %5s %9.2f %6s'%(r['Name'].replace('(anonymous namespace)::','').split('(')[0][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
