#!/bin/bash
# chunk width of the streaming kernels (APA_M1S_PIX, development build only) with the prefetch where it belongs
# (round 6): headline fp32, eval, N = 512, bf16 rank-1 head, cfg 003
out=${1:-gpurun_out/r06_pix}; mkdir -p $out
L=attentionalpoolingaction_amd/custom_ops/libapa_hip_ablate.so
for rep in 1 2; do
for pix in 2 4 8 1; do
  APA_LIB_PATH=$L APA_M1S_PIX=$pix python3 bench.py --gpus 1 --no-cpu-baseline --steps 200 --warmup 20 \
     --extra-only cfg002_eval,cfg003_bf16_train,hmdb51_rank1_bf16_train,cfg002_train_n512 > $out/pix${pix}_$rep.json 2> $out/pix${pix}_$rep.err
done; done
python3 - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], 'pix*.json'))):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unreadable'); continue
    row = ['%s headline %.2f us fwd %.2f bwd %.2f' % (os.path.basename(f)[:-5], d['ms_per_step']*1e3, d['roofline_fwd']['kernel_avg_us'], d['roofline']['kernel_avg_us'])]
    for k, v in d.get('extra', {}).items():
        if 'ms_per_step' in v:
            row.append('%s %.2f' % (k.replace('cfg002_','').replace('_bf16_train',''), v['ms_per_step']*1e3))
            if 'roofline_fwd' in v: row.append('(fwd %.1f bwd %.1f)' % (v['roofline_fwd']['kernel_avg_us'], v['roofline']['kernel_avg_us']))
    print(' | '.join(row))
PY
