#!/bin/bash
# A/B of two builds of the library on ONE box, interleaved (A B A B) so that box drift hits both arms alike:
#   tools/ab_libs.sh <out_dir> <libA.so> <libB.so> [bench.py args...]
# Prints, per arm and run, the headline and the selected `extra` lines.
out=$1; A=$2; B=$3; shift 3
mkdir -p "$out"
for r in 1 2; do
  for arm in A B; do
    lib=$A; [ $arm = B ] && lib=$B
    APA_LIB_PATH=$lib python3 bench.py --gpus 1 --no-cpu-baseline "$@" > "$out/${arm}_$r.json" 2> "$out/${arm}_$r.err"
  done
done
python3 - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, '[AB]_*.json'))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'unreadable', e); continue
    row = ['%s headline %.2f us fwd %.2f bwd %.2f' % (os.path.basename(f)[:-5], d['ms_per_step'] * 1e3,
           d['roofline_fwd']['kernel_avg_us'], d['roofline']['kernel_avg_us'])]
    for k, v in d.get('extra', {}).items():
        if 'ms_per_step' in v:
            row.append('%s %.2f' % (k, v['ms_per_step'] * 1e3))
            if 'roofline_fwd' in v:
                row.append('(fwd %.1f bwd %.1f)' % (v['roofline_fwd']['kernel_avg_us'], v['roofline']['kernel_avg_us']))
        elif 'error' in v:
            row.append('%s ERR %s' % (k, v['error'][:60]))
    print(' | '.join(row))
PY
