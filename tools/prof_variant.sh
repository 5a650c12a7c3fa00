#!/bin/bash
# one rocprofv3 --kernel-trace --stats run of a tools/bench_dense.py workload under a set of A/B knobs
# (development library, make ABLATE=1): prof_variant.sh <name> "<bench_dense args>" [ENV=VAL ...]
name=$1; shift; args=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/var_$name; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env APA_LIB_PATH=$R/attentionalpoolingaction_amd/custom_ops/libapa_hip_ablate.so "$@" \
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/bench_dense.py $args > $O/bench.log 2>&1
python - "$O" "$name" <<'PY'
import csv, glob, json, os, sys
O, name = sys.argv[1:3]
try:
    b = json.loads([l for l in open(os.path.join(O, 'bench.log')) if l.startswith('{"workload"')][-1])
    print('== %s: %.1f us/step under the profiler' % (name, b['ms_per_step'] * 1e3))
except Exception as e:
    print('== %s: no bench line (%s)' % (name, e)); print(open(os.path.join(O, 'bench.log')).read()[-1500:])
f = glob.glob(os.path.join(O, '**', '*kernel_stats.csv'), recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if int(r['Calls']) >= 50]
    print('   launches/step: %.1f' % (sum(int(r['Calls']) for r in rows) / 255.0))
    for r in rows:
        n = r['Name'].replace('void apa::', '').replace('apa::', '').replace('(anonymous namespace)::', '').split('(')[0][:70]
        print('   %-72s %4s x %7.2f us' % (n, r['Calls'], float(r['AverageNs']) / 1e3))
PY
