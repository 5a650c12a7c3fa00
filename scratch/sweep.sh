#!/bin/bash
# quick perf sweeps (one JSON line each -> compact summary)
summ() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], 'img/s', d['ms_per_step']*1e3, 'us/step  bwd_main', r['kernel_avg_us'], 'us', r['achieved'], 'GB/s frac', r['frac'], 'step_frac', d['step_roofline_frac'])"; }
for t in 256 384 512 640; do APA_M1_TARGET_BLOCKS=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | summ "N=32 target=$t"; done
for n in 16 64 128 256 512; do python bench.py --steps 100 --warmup 10 --batch $n --no-cpu-baseline 2>/dev/null | summ "N=$n"; done
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --eval-mode 2>/dev/null | summ "N=32 eval"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --dtype bf16 2>/dev/null | summ "N=32 bf16"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dtype bf16 --batch 256 2>/dev/null | summ "N=256 bf16"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --softmax-att 2>/dev/null | summ "N=32 softmax"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --graph 2>/dev/null | summ "N=32 graph"
