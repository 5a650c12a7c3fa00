#!/bin/bash
run() { python bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %9.0f img/s %7.2f us/step  k=%.2f' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_avg_us']))" "$@"; }
for i in 1 2; do
run
run --per-op-calls
run --graph
run --force-dist
run --force-dist --overlap off
done
