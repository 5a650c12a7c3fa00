#!/bin/bash
summ() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], 'img/s', round(d['ms_per_step']*1e3,2), 'us/step  bwd_main', r['kernel_avg_us'], 'us frac', r['frac'])"; }
for pix in 1 2 4 8; do for t in 512 1024; do
APA_M1S_PIX=$pix APA_M1_TARGET_BLOCKS=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | summ "pix=$pix target=$t"
done; done
APA_M1_STREAM=0 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | summ "old"
