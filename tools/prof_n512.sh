#!/bin/bash
# per-kernel durations of the headline step at per-GPU batch 512 (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/prof_n512_$1; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
shift
env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --batch 512 --rotate 2 --steps 40 --warmup 5 --no-extra --no-cpu-baseline > $O/bench.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, json, os, sys
O = sys.argv[1]
try:
    b = json.loads([l for l in open(os.path.join(O, 'bench.log')) if l.startswith('{"metric"')][-1])
    print('== N=512: %.1f us/step under the profiler' % (b['ms_per_step'] * 1e3))
except Exception as e:
    print('no bench line', e); print(open(os.path.join(O, 'bench.log')).read()[-800:])
f = glob.glob(os.path.join(O, '**', '*kernel_stats.csv'), recursive=True)
for r in csv.DictReader(open(f[0])):
    if int(r['Calls']) >= 40:
        print('   %-70s %5s x %8.2f us' % (r['Name'].replace('void apa::', '').split('(')[0][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
