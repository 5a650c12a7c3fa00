#!/bin/bash
# usage: prof.sh <tag> [bench args...]  -> prints per-kernel stats
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $out/bench.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
echo "== $tag: $(tail -1 $out/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']*1e3)")"
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    n=r['Name'][:70]; c=int(r['Calls']); avg=float(r['AverageNs'])/1e3
    if c>=200: print('  %-70s %5d %8.2f us'%(n,c,avg)); tot+=avg if c>=200 and c<=330 else 0
print('  sum of per-step kernels: %.2f us'%tot)
PY
