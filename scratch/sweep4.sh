#!/bin/bash
summ() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], 'img/s', round(d['ms_per_step']*1e3,2), 'us/step  bwd_main', r['kernel_avg_us'], 'us frac', r['frac'], 'step_frac', d['step_roofline_frac'])"; }
for pix in 1 4 8; do APA_M1S_PIX=$pix python bench.py --steps 300 --warmup 30 --no-cpu-baseline --dtype bf16 2>/dev/null | summ "N=32 bf16 pix=$pix"; done
for pix in 4 8; do APA_M1S_PIX=$pix python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dtype bf16 --batch 256 2>/dev/null | summ "N=256 bf16 pix=$pix"; done
for t in 1024 2048 4096; do APA_M1_TARGET_BLOCKS=$t python bench.py --steps 50 --warmup 5 --batch 512 --no-cpu-baseline 2>/dev/null | summ "N=512 f32 target=$t"; done
for pix in 1 4; do APA_M1S_PIX=$pix APA_M1_TARGET_BLOCKS=2048 python bench.py --steps 50 --warmup 5 --batch 512 --no-cpu-baseline 2>/dev/null | summ "N=512 f32 target=2048 pix=$pix"; done
