#!/bin/bash
# Experiment behind DESIGN.md section 5 (host side): host time to enqueue a step vs wall time per step, one-call
# step vs the three per-op calls vs the N > 1 flows (short runs so the launch queue never fills).
for a in "" "--per-op-calls" "--force-dist" "--force-dist --overlap off"; do
echo "== $a"; python bench.py --steps 30 --warmup 30 --no-cpu-baseline $a 2>&1 >/dev/null | grep "host enq"
python bench.py --steps 30 --warmup 30 --no-cpu-baseline $a 2>&1 >/dev/null | grep "host enq"
done
