#!/bin/bash
# End-to-end evidence (run through gpurun): tools/bench_e2e.py for BASELINE configs[1] / [2]
#   1. plain runs (the JSON lines: img/s + the head's share by HIP events)
#   2. rocprofv3 --kernel-trace --stats of each (where the step's time goes, kernel by kernel)
#   3. rocprofv3 --pmc FETCH_SIZE restricted to the head's kernels: HBM-side bytes the head's forward streaming
#      kernel fetches INSIDE the step (conv5 was written by block4 microseconds earlier: Infinity Cache or HBM?)
#      against the 51.4 MB it reads in the rotating-buffer bench (gfx950: FETCH_SIZE reports half of a wide
#      coalesced stream -> doubled, the convention of tools/profile_summarise.py)
# -> gpurun_out/profiles_out/<tag>_e2e_summary.md      Usage: profile_e2e.sh <tag>
tag=${1:-r04}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/profiles_out
O=$R/gpurun_out/prof_${tag}_e2e; rm -rf $O; mkdir -p $O
cd $R
for wl in eval002 train003; do
  python tools/bench_e2e.py --workload $wl > $O/$wl.json 2> $O/$wl.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python tools/bench_e2e.py --workload $wl --steps 3 --warmup 2 > $O/stats_$wl.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "apa::" --output-format csv -d $O/pmc_$wl -- python tools/bench_e2e.py --workload $wl --steps 2 --warmup 1 > $O/pmc_$wl.log 2>&1
done
python tools/bench_e2e.py --workload eval002 --fuse-final-relu > $O/eval002_fused_relu.json 2> $O/eval002_fused_relu.err
python - "$O" "$R/gpurun_out/profiles_out/${tag}_e2e_summary.md" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
O, dst = sys.argv[1:3]
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:90]
out = ['# end-to-end steps (tools/bench_e2e.py, one MI355X): backbone + head, where the time goes\n']
for wl in ('eval002', 'train003', 'eval002_fused_relu'):
    try:
        b = json.loads([l for l in open(os.path.join(O, wl + '.json')) if l.startswith('{')][-1])
    except Exception as e:
        out.append('\n## %s: no result (%s)\n' % (wl, e)); continue
    hs = b['head_share']
    out.append('\n## %s\n\n%s\n\n**%.1f img/s, %.2f ms/step**; head inside the step: forward %.3f ms, backward %.3f ms = **%.2f %% of the step** (%s)\n'
               % (wl, b['workload'], b['images_per_sec'], b['ms_per_step'], hs['head_fwd_ms'], hs['head_bwd_ms'],
                  100.0 * (hs['fraction_of_step'] or 0), hs['how']))
    f = glob.glob(os.path.join(O, 'stats_' + wl, '**', '*kernel_stats.csv'), recursive=True)
    if f:
        rows = list(csv.DictReader(open(f[0])))
        tot = sum(float(r['TotalDurationNs']) for r in rows)
        apa = sum(float(r['TotalDurationNs']) for r in rows if 'apa::' in r['Name'])
        out.append('\nrocprofv3 --kernel-trace --stats (3 timed + 2 warm-up + 4 probe steps): library kernels (`apa::`) = %.2f %% of all kernel time\n\n'
                   '| kernel | calls | avg us | %% of kernel time |\n|---|---|---|---|\n' % (100.0 * apa / max(tot, 1)))
        for r in rows[:14]:
            out.append('| `%s` | %s | %.1f | %s |\n' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
        for r in rows[14:]:
            if 'apa::' in r['Name'] and float(r['AverageNs']) > 3000:
                out.append('| `%s` | %s | %.1f | %s |\n' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
    f = glob.glob(os.path.join(O, 'pmc_' + wl, '**', '*counter_collection.csv'), recursive=True)
    if f:
        acc = defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r['Counter_Name'] == 'FETCH_SIZE':
                acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
        out.append('\nHBM-side fetch per launch of the library kernels INSIDE this step (rocprofv3 --pmc FETCH_SIZE, KB -> MB, x2: '
                   'gfx950 counts 64 B per 128-B request of a wide stream):\n\n| kernel | launches | fetched MB / launch |\n|---|---|---|\n')
        for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
            out.append('| `%s` | %d | %.2f |\n' % (n, len(v), 2 * sum(v) / len(v) * 1024 / 1e6))
open(dst, 'w').write(''.join(out))
print(''.join(out))
PY
