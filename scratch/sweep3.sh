#!/bin/bash
summ() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], 'img/s', round(d['ms_per_step']*1e3,2), 'us/step  bwd_main', r['kernel_avg_us'], 'us frac', r['frac'], 'step_frac', d['step_roofline_frac'])"; }
for n in 16 64 128 512; do python bench.py --steps 100 --warmup 10 --batch $n --no-cpu-baseline 2>/dev/null | summ "N=$n f32"; done
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --dtype bf16 2>/dev/null | summ "N=32 bf16"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dtype bf16 --batch 256 2>/dev/null | summ "N=256 bf16"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --softmax-att 2>/dev/null | summ "N=32 softmax"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --eval-mode 2>/dev/null | summ "N=32 eval"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --graph 2>/dev/null | summ "N=32 graph"
for pix in 1 4; do APA_M1S_PIX=$pix python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | summ "N=32 pix=$pix"; done
for t in 384 768; do APA_M1_TARGET_BLOCKS=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | summ "N=32 target=$t"; done
APA_M1_STREAM=0 python bench.py --steps 100 --warmup 10 --batch 512 --no-cpu-baseline 2>/dev/null | summ "N=512 f32 old-stream"
