#!/usr/bin/env python
"""Turn what tools/gpu_r02_final.sh left under gpurun_out/ into the committed evidence under profiles/:
the headline summaries (tools/profile_summarise.py on prof_<tag>, prof_<tag>_n128, prof_<tag>_n512), the
summaries of the tools/bench_dense.py workloads and the default bench line.  Usage: profile_collect.py <tag>"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(d, pat):
    fs = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def dense_summary(tag, name, args):
    d = os.path.join(ROOT, 'gpurun_out', 'prof_{}_{}'.format(tag, name))
    stats = newest(d, '*kernel_stats.csv')
    if stats is None:
        print('skip', name)
        return
    rows = list(csv.DictReader(open(stats)))
    line = [l for l in open(os.path.join(d, 'bench.log')) if l.startswith('{"workload"')][-1]
    b = json.loads(line)
    r = b['roofline']
    out = os.path.join(ROOT, 'profiles', '{}_{}_summary.md'.format(tag, name))
    with open(out, 'w') as o:
        o.write('# {}_{} -- rocprofv3 --kernel-trace --stats of `python tools/bench_dense.py {}` (MI355X, gfx950)\n\n'
                .format(tag, name, args))
        o.write('{}\n\nunder the profiler: {:.0f} img/s, {:.4f} ms/step (whole-step roofline fraction {}: {} {} of {})\n\n'
                .format(b['workload'], b['images_per_sec'], b['ms_per_step'], r['frac'], r['achieved'], r['unit'], r['peak']))
        o.write('| kernel | calls | avg us | % |\n|---|---|---|---|\n')
        for x in rows:
            if int(x['Calls']) >= 50 and float(x['Percentage']) >= 0.5:
                o.write('| `{}` | {} | {:.2f} | {} |\n'.format(
                    x['Name'].replace('void apa::', '').replace('apa::', '').replace('(anonymous namespace)::', '')
                    .split('(')[0][:100], x['Calls'], float(x['AverageNs']) / 1e3, x['Percentage']))
    print('wrote', out)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
    go = os.path.join(ROOT, 'gpurun_out')
    py = [sys.executable, os.path.join(ROOT, 'tools', 'profile_summarise.py')]
    subprocess.check_call(py + [os.path.join(go, 'prof_' + tag), tag], stdout=subprocess.DEVNULL)
    for n in (128, 512):
        d = os.path.join(go, 'prof_{}_n{}'.format(tag, n))
        if os.path.isdir(d):
            subprocess.check_call(py + [d, '{}_n{}'.format(tag, n), '--batch', str(n)], stdout=subprocess.DEVNULL)
    dense_summary(tag, 'cfg003', '--workload cfg003')
    dense_summary(tag, 'perclass', '--workload perclass')
    dense_summary(tag, 'perclass393', '--workload perclass --classes 393')
    dense_summary(tag, 'eval002', '--workload eval002')
    src = os.path.join(go, 'bench_final.json')
    if os.path.exists(src):
        line = [l for l in open(src) if l.startswith('{"metric"')][-1]
        open(os.path.join(ROOT, 'profiles', tag + '_bench_line.json'), 'w').write(line)
        print('bench line:', line[:160])


if __name__ == '__main__':
    main()
