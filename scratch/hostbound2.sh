#!/bin/bash
for a in "" "--per-op-calls" "--force-dist" "--force-dist --overlap off"; do
echo "== $a"; python bench.py --steps 30 --warmup 30 --no-cpu-baseline $a 2>&1 >/dev/null | grep "host enq"
python bench.py --steps 30 --warmup 30 --no-cpu-baseline $a 2>&1 >/dev/null | grep "host enq"
done
