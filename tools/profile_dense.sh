#!/bin/bash
# rocprofv3 --kernel-trace --stats of the MFMA-bound workloads (tools/bench_dense.py) -> profiles/<tag>_<workload>_*.
# Usage (through gpurun): tools/profile_dense.sh <tag>
tag=${1:-r01}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/profiles
for wl in cfg003 perclass "perclass --classes 393" eval002 rank1; do
  name=$(echo $wl | sed 's/ --classes //')
  O=$R/gpurun_out/prof_${tag}_$name; rm -rf $O; mkdir -p $O
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/bench_dense.py --workload $wl > $O/bench.log 2>&1
  python - "$O" "$R/profiles/${tag}_${name}" "$wl" <<'PY'
import csv, glob, json, os, sys
out_dir, prefix, wl = sys.argv[1:4]
f = glob.glob(os.path.join(out_dir, '**', '*kernel_stats.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
line = [l for l in open(os.path.join(out_dir, 'bench.log')) if l.startswith('{"workload"')][-1]
b = json.loads(line)
with open(prefix + '_summary.md', 'w') as o:
    o.write('# {} -- rocprofv3 --kernel-trace --stats of `python tools/bench_dense.py --workload {}` (MI355X)\n\n'.format(os.path.basename(prefix), wl))
    o.write('{}\n\nunder the profiler: {:.0f} img/s, {:.3f} ms/step\n\n'.format(b['workload'], b['images_per_sec'], b['ms_per_step']))
    o.write('| kernel | calls | avg us | % |\n|---|---|---|---|\n')
    for r in rows:
        if int(r['Calls']) >= 50 and float(r['Percentage']) >= 0.5:
            o.write('| `{}` | {} | {:.2f} | {} |\n'.format(r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:100], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
print(open(prefix + '_summary.md').read())
PY
  cp $R/profiles/${tag}_${name}_summary.md $R/gpurun_out/ 2>/dev/null   # gpurun merges gpurun_out/ back, not profiles/
done
